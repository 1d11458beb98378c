from .face_exception import FaceRecognitionException  # noqa: F401

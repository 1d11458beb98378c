class FaceRecognitionException(BaseException):
    """Error type of the face pipeline; derives from BaseException exactly as the
    reference's does (deep_insight_face/exceptions/face_exception.py:1), so handlers
    written against the reference keep catching it."""

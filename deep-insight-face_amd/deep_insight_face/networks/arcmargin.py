"""ArcMargin (ArcFace) logits head on the MI355X.  Absent from the reference (its losses
are triplet / contrastive only: networks/triplet.py:16-46, networks/siamese.py:32-39);
named by north_star.  logits = s * cos(theta + m * onehot(label)), Deng et al. 2019."""
import ctypes

import torch

from .. import _native as N


class ArcMarginHead:
    def __init__(self, class_centres, s=64.0, m=0.5):
        self._dev = N.require_device()
        w, _ = N.to_device_f32(class_centres, self._dev)
        if w.dim() != 2:
            raise ValueError('class centres must be [C, d]')
        self.n_classes, self.emd_size = int(w.shape[0]), int(w.shape[1])
        self._h = ctypes.c_void_p()
        N.check(N.lib.dif_arcmargin_create(ctypes.byref(self._h), self.emd_size, self.n_classes, float(s), float(m)),
                ValueError)
        N.check(N.lib.dif_arcmargin_set_weight(self._h, N.ptr(w), N.stream_ptr()))
        torch.cuda.current_stream().synchronize()

    def logits(self, embeddings, labels=None):
        e, was_np = N.to_device_f32(embeddings, self._dev)
        if e.dim() != 2 or e.shape[1] != self.emd_size:
            raise ValueError('embeddings must be [B, %d]' % self.emd_size)
        B = e.shape[0]
        out = torch.empty((B, self.n_classes), dtype=torch.float32, device=self._dev)
        lab = None
        if labels is not None:
            lab = torch.as_tensor(labels).to(device=self._dev, dtype=torch.int64).contiguous()
            if lab.shape != (B,):
                raise ValueError('labels must be [B]')
        if B:
            N.check(N.lib.dif_arcmargin_logits(self._h, N.ptr(e), N.ptr(lab) if lab is not None else None, B,
                                               N.ptr(out), N.stream_ptr()))
        return out.cpu().numpy() if was_np else out

    def close(self):
        if self._h:
            N.lib.dif_arcmargin_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

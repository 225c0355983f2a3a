"""GPU clip preprocessing: the evaluation branch of the reference's VideoDataset.__getitem__
(video_dataset/dataset.py:117-139) from decoded uint8 frames to the model input, on the device.

The reference does this on the CPU inside DataLoader workers: float conversion of EVERY decoded frame,
normalisation, `F.interpolate(bilinear)`, centre crop, and only then the temporal crop.  Here the uint8
frames go to the GPU as they are (4x fewer PCIe / HBM bytes than fp32) and one HBM-bound kernel produces
the (3, T, size, size) clip, touching only the T frames the temporal crop keeps.  Decoding (PyAV) and the
training-time augmentations (auto-augment, random resized crop, dataset.py:97-115) stay on the host.
"""
import torch

from . import hip

# the statistics every eval/train script passes (eval_scripts/eval_updrs.sh:9-10, k400_eval.sh:14-15)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ClipPreprocessor:
    """Mirror of VideoDataset(random_sample=False, ...) (the evaluation branch, any view counts upstream accepts).

    Same argument names as the reference constructor (dataset.py:23-33) for the ones that matter here.
    """

    def __init__(self, num_frames=8, sampling_rate=1, spatial_size=224, mean=CLIP_MEAN, std=CLIP_STD,
                 num_spatial_views=1, num_temporal_views=1):
        if num_spatial_views not in (1, 3):
            raise NotImplementedError()          # dataset.py:201-202
        # upstream builds all num_spatial_views x num_temporal_views crops but returns only the first one
        # (`frames = frames[0]`, dataset.py:134-139): the top/left spatial crop and the temporal crop starting at frame 0
        self.num_spatial_views, self.num_temporal_views = num_spatial_views, num_temporal_views
        self.num_frames, self.sampling_rate, self.spatial_size = num_frames, sampling_rate, spatial_size
        self.mean = tuple(float(v) for v in torch.as_tensor(mean).flatten().tolist())
        self.std = tuple(float(v) for v in torch.as_tensor(std).flatten().tolist())
        self._lut = {}

    def lut(self, device):
        """(v/255 - mean) / std for the 256 byte values per channel, the reference's expression evaluated once on the host"""
        if device not in self._lut:
            self._lut[device] = hip.clip_lut(self.mean, self.std, device)
        return self._lut[device]

    def check(self, videos):
        for v in videos:
            self._check(v)

    def _check(self, frames):
        if not frames.is_cuda:
            raise hip.GavaError("ClipPreprocessor takes device tensors (no CPU fallback)")
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [n_frames, H, W, 3] (to_rgb().to_ndarray() order)")
        h, w = frames.shape[1], frames.shape[2]
        s = self.spatial_size
        new_h, new_w = (s, w * s // h) if h < w else (h * s // w, s)
        if self.num_spatial_views == 1:
            assert min(new_h, new_w) >= s   # dataset.py:182
        else:
            assert min(new_h, new_w) == s   # dataset.py:189

    def __call__(self, frames, out=None):
        """frames: uint8 [n, H, W, 3] on the GPU -> fp32 [3, T, S, S] (dataset.py returns frames[0] of this shape)."""
        self._check(frames)
        T, S = self.num_frames, self.spatial_size
        if out is None:
            out = torch.empty(3, T, S, S, dtype=torch.float32, device=frames.device)
        hip.preprocess_clip(frames.contiguous(), out, T=T, rate=self.sampling_rate, size=S, mean=self.mean, std=self.std,
                            first_temporal_view=self.num_temporal_views > 1, first_spatial_view=self.num_spatial_views == 3,
                            lut=self.lut(frames.device))
        return out

    def batch(self, videos):
        """list of uint8 [n_i, H_i, W_i, 3] -> fp32 [B, 3, T, S, S], each clip written in place."""
        T, S = self.num_frames, self.spatial_size
        x = torch.empty(len(videos), 3, T, S, S, dtype=torch.float32, device=videos[0].device)
        for b, v in enumerate(videos):
            self(v, out=x[b])
        return x

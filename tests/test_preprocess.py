"""Clip preprocessing: HIP kernel vs the CPU restatement of video_dataset/dataset.py:117-139."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as po

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def _video(n, h, w, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8))


def test_oracle_matches_plain_loops():
    """The torch restatement against scalar Python arithmetic on a tiny case (pins the crop/resize bookkeeping)."""
    v = _video(5, 6, 9, 0)
    S, T, rate = 4, 3, 2
    got = po.preprocess_clip(v, T, rate, S, MEAN, STD)
    assert got.shape == (3, T, S, S)
    h, w = 6, 9
    new_h, new_w = S, w * S // h            # 4, 6
    h_st, w_st = (new_h - S) // 2, (new_w - S) // 2
    seg = (T - 1) * rate + 1
    st = max(5 - seg, 0) // 2
    for t in range(T):
        f = min(st + t * rate, 4)
        for c in range(3):
            for y in range(S):
                for x in range(S):
                    sy = max(np.float32(h / new_h) * np.float32(y + h_st + 0.5) - np.float32(0.5), 0)
                    sx = max(np.float32(w / new_w) * np.float32(x + w_st + 0.5) - np.float32(0.5), 0)
                    y0, x0 = int(sy), int(sx)
                    y1, x1 = min(y0 + 1, h - 1), min(x0 + 1, w - 1)
                    ly, lx = float(sy) - y0, float(sx) - x0
                    val = lambda yy, xx: (float(v[f, yy, xx, c]) / 255. - MEAN[c]) / STD[c]
                    ref = (1 - ly) * ((1 - lx) * val(y0, x0) + lx * val(y0, x1)) + ly * ((1 - lx) * val(y1, x0) + lx * val(y1, x1))
                    assert abs(float(got[c, t, y, x]) - ref) < 2e-5


def test_short_video_repeats_last_frame():
    v = _video(3, 8, 8, 1)
    out = po.preprocess_clip(v, 4, 2, 8, MEAN, STD)    # seg_len 7 > 3 frames: indices 0, 2, 2(pad), 2(pad)
    assert torch.equal(out[:, 2], out[:, 1]) and torch.equal(out[:, 3], out[:, 1])


CASES = [  # n_frames, H, W, T, rate, size
    (20, 240, 320, 8, 2, 224),     # landscape, the usual UPDRS/K400 shape class
    (9, 320, 240, 8, 1, 224),      # portrait
    (5, 256, 256, 8, 1, 224),      # square, short video (last frame repeated)
    (40, 360, 640, 16, 2, 224),    # 16 frames
    (12, 224, 224, 8, 1, 224),     # no resize at all: exact normalisation only
    (10, 181, 333, 4, 3, 96),      # odd sizes
]


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,T,rate,size", CASES)
def test_gpu_preprocess_matches_oracle(n, h, w, T, rate, size):
    from gava_clip_amd.preprocess import ClipPreprocessor
    v = _video(n, h, w, n * 1000 + h)
    ref = po.preprocess_clip(v, T, rate, size, MEAN, STD)
    pre = ClipPreprocessor(num_frames=T, sampling_rate=rate, spatial_size=size, mean=MEAN, std=STD)
    got = pre(v.cuda()).cpu()
    assert got.shape == ref.shape
    # fp32 arithmetic in the same order as torch's kernel; tolerance: 2e-6 of the value range (|x| < 2.7)
    err = (got - ref).abs().max().item()
    assert err <= 6e-6, err


@pytest.mark.gpu
def test_gpu_preprocess_batch_writes_in_place():
    from gava_clip_amd.preprocess import ClipPreprocessor
    pre = ClipPreprocessor(num_frames=8, sampling_rate=1, spatial_size=224)
    vids = [_video(10, 240, 320, 7), _video(8, 300, 260, 8)]
    x = pre.batch([v.cuda() for v in vids])
    assert x.shape == (2, 3, 8, 224, 224)
    for b, v in enumerate(vids):
        ref = po.preprocess_clip(v, 8, 1, 224, MEAN, STD)
        assert (x[b].cpu() - ref).abs().max().item() <= 6e-6


@pytest.mark.gpu
def test_gpu_preprocess_rejects_host_tensor_and_small_frames():
    from gava_clip_amd import hip
    from gava_clip_amd.preprocess import ClipPreprocessor
    pre = ClipPreprocessor()
    with pytest.raises(hip.GavaError):
        pre(_video(8, 240, 320, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,views", [(30, 240, 320, (1, 10)), (30, 320, 240, (3, 10)), (12, 224, 400, (3, 1))])
def test_gpu_preprocess_first_of_many_views(n, h, w, views):
    """evaluate.py's defaults (--num_temporal_views 10): upstream builds every crop and returns the first."""
    from gava_clip_amd.preprocess import ClipPreprocessor
    v = _video(n, h, w, n + h)
    ref = po.preprocess_clip(v, 8, 2, 224, MEAN, STD, num_spatial_views=views[0], num_temporal_views=views[1])
    pre = ClipPreprocessor(num_frames=8, sampling_rate=2, spatial_size=224, mean=MEAN, std=STD,
                           num_spatial_views=views[0], num_temporal_views=views[1])
    got = pre(v.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 6e-6

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


def test_oracle_reproduces_the_reference_dataset_bit_for_bit(golden_dir):
    """tests/golden/preprocess_ref.npz holds what the REFERENCE's own VideoDataset.__getitem__ (evaluation branch,
    video_dataset/dataset.py:78-139,163-200) returns for nine synthetic decoded videos (tools/gen_golden.py --preprocess: PyAV and
    torchvision replaced by stand-ins, everything from `to_rgb().to_ndarray()` on is upstream's code), as the sha256 of the fp32
    bytes and a strided sample.  The restatement must reproduce every byte: this pins the checker of the GPU input path
     to the reference."""
    import hashlib
    import os
    g = np.load(os.path.join(golden_dir, "preprocess_ref.npz"))
    for i, (n, h, w, T, rate, size, sv, tv) in enumerate(g["cases"].tolist()):
        seed = n * 1000 + h if sv == 1 and tv == 1 else n + h
        v = _video(n, h, w, seed)
        a = po.preprocess_clip(v, T, rate, size, MEAN, STD, num_spatial_views=sv, num_temporal_views=tv).contiguous().numpy()
        assert np.array_equal(a.reshape(-1)[::max(1, a.size // 4096)][:4096], g[f"sample_{i}"]), i
        assert hashlib.sha256(a.tobytes()).digest() == g[f"sha256_{i}"].tobytes(), i


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


# ---- the fused input path: the patch-embedding GEMM reads the decoded uint8 frames itself ------------

@pytest.mark.gpu
@pytest.mark.parametrize("size,patch,T,rate,shapes", [
    (64, 16, 4, 2, [(9, 80, 120), (5, 100, 70), (12, 64, 64)]),     # P % 8 == 0
    (56, 14, 2, 1, [(3, 60, 90), (2, 57, 56)]),                      # ViT-L/14's patch: K = 588 padded to 640
])
def test_patch_embedding_reads_uint8_frames(size, patch, T, rate, shapes):
    """A tiles built from the decoded uint8 videos (temporal crop, normalisation, bilinear resize, centre crop per loaded
    pixel) == the two-step form (gava_preprocess_clip writes the fp32 clip, the im2col-free GEMM reads it), bit for bit;
    videos of different sizes in one batch."""
    from gava_clip_amd import hip
    from gava_clip_amd.preprocess import ClipPreprocessor
    d = torch.device("cuda")
    vids = [_video(n, h, w, 10 + i).to(d) for i, (n, h, w) in enumerate(shapes)]
    pre = ClipPreprocessor(num_frames=T, sampling_rate=rate, spatial_size=size, mean=MEAN, std=STD)
    x = pre.batch(vids)
    B, D = len(vids), 256
    g = size // patch
    n, K = g * g, 3 * patch * patch
    Kp = (K + 63) // 64 * 64
    gen = torch.Generator().manual_seed(3)
    W = torch.zeros(D, Kp)
    W[:, :K] = torch.randn(D, K, generator=gen) * K ** -0.5
    W16 = W.to(d).half()
    bias, pos, tim = (torch.randn(s_, generator=gen).to(d) for s_ in ((D,), (n + 1, D), (T, D)))
    outs = []
    desc, keep = hip.clip_descriptors(vids, T=T, rate=rate, size=size)
    for kw in (dict(frames=x), dict(clips=desc, clip_lut=pre.lut(d))):
        X = torch.zeros(B * T * (n + 1), D, device=d)
        hip.gemm(None, W16, bias, X, epilogue=hip.EPI_F32_PATCH, prec=hip.PREC_F16, pos=pos, time=tim, n_patches=n, T=T,
                 M=B * T * n, frame_size=size, patch=patch, **kw)
        outs.append(X)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[0].abs().max()) > 0.1
    # the form the forward uses: 16-bit patch matrix first (gava_patchify), then the GEMM with an ordinary A operand
    ref = x.view(B, 3, T, g, patch, g, patch).permute(0, 2, 3, 5, 1, 4, 6).reshape(B * T * n, K).half()   # unfold, RN to fp16
    for kw in (dict(x=x), dict(clips=desc, clip_lut=pre.lut(d))):
        A = torch.full((B * T * n, Kp), 7.0, device=d, dtype=torch.float16)
        hip.patchify(A, B=B, T=T, size=size, patch=patch, prec=hip.PREC_F16, **kw)
        assert torch.equal(A[:, :K], ref) and not A[:, K:].any()
        X = torch.zeros(B * T * (n + 1), D, device=d)
        hip.gemm(A, W16, bias, X, epilogue=hip.EPI_F32_PATCH, prec=hip.PREC_F16, pos=pos, time=tim, n_patches=n, T=T, M=B * T * n)
        assert torch.equal(X, outs[0])
    with pytest.raises(hip.GavaError):
        hip.patchify(A, B=B, T=T, size=size, patch=patch, prec=hip.PREC_F16)                      # no source
    with pytest.raises(hip.GavaError):
        hip.patchify(torch.empty(B * T * n, K - 8, device=d, dtype=torch.float16), B=B, T=T, size=size, patch=patch,
                     prec=hip.PREC_F16, x=x)                                                      # rows too short for 3 P P


@pytest.mark.gpu
def test_forward_frames_equals_forward_of_the_preprocessed_batch():
    """VitaCLIP.forward_frames(videos, preprocessor) == forward(preprocessor.batch(videos)) bit for bit, and both meet the
    oracle pipeline (CPU restatement of dataset.py:117-139 followed by the forward oracle) within 1e-3."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import CLASSES_3, model_kwargs, synth_torch_state, rel_to_max
    from gava_clip_amd import VitaCLIP, hip
    from gava_clip_amd.config import TINY
    from gava_clip_amd.preprocess import ClipPreprocessor
    from oracle.vita_oracle import Oracle
    sd = synth_torch_state(TINY, 3)
    m = VitaCLIP(**model_kwargs(TINY, CLASSES_3))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    vids_cpu = [_video(11, 90, 130, 21), _video(6, 120, 80, 22)]
    vids = [v.cuda() for v in vids_cpu]
    pre = ClipPreprocessor(num_frames=TINY.num_frames, sampling_rate=2, spatial_size=TINY.input_size, mean=MEAN, std=STD)
    with torch.no_grad():
        a = m.forward_frames(vids, pre)[0]
        b = m(pre.batch(vids))[0]
    assert torch.equal(a, b)
    x_ref = torch.stack([po.preprocess_clip(v, TINY.num_frames, 2, TINY.input_size, MEAN, STD) for v in vids_cpu])
    want = Oracle(TINY, sd, torch.cat(m.tokenized_prompts)).forward(x_ref)["logits"].numpy()
    assert rel_to_max(a.cpu().numpy(), want) < 1e-3
    with pytest.raises(hip.GavaError):
        m.forward_frames(vids, pre)            # grad enabled: the evaluation data path only

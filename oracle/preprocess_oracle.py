"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's evaluation preprocessing.

Follows video_dataset/dataset.py:117-139 (+ :163-186) step by step with the same torch calls, on the CPU:
float()/255, (x-mean)/std, permute to (C,T,H,W), F.interpolate(bilinear, align_corners=False) of ALL frames to the
short-side size, centre crop, temporal crop.  The arithmetic inside F.interpolate is third-party (torch).  PINNED since
round 3: tools/gen_golden.py --preprocess runs the reference's own VideoDataset.__getitem__ (its module imported with
stand-ins for PyAV - a container that yields synthetic frames - and for torchvision, neither of which the evaluation
branch uses after decoding) on nine synthetic videos; tests/test_preprocess.py checks that this restatement reproduces
those outputs BIT FOR BIT (sha256 of the fp32 bytes, tests/golden/preprocess_ref.npz).  Imported by tests/ only; the
product (gava_clip_amd/preprocess.py) never touches it.
"""
import torch


def preprocess_clip(frames_u8, num_frames, sampling_rate, spatial_size, mean, std, num_spatial_views=1,
                    num_temporal_views=1):
    mean = torch.as_tensor(mean, dtype=torch.float32)
    std = torch.as_tensor(std, dtype=torch.float32)
    frames = torch.as_tensor(frames_u8).float() / 255.                       # :118-119
    frames = (frames - mean) / std                                           # :121
    frames = frames.permute(3, 0, 1, 2)                                      # :122  C, T, H, W
    if frames.size(-2) < frames.size(-1):                                    # :124-129
        new_width = frames.size(-1) * spatial_size // frames.size(-2)
        new_height = spatial_size
    else:
        new_height = frames.size(-2) * spatial_size // frames.size(-1)
        new_width = spatial_size
    frames = torch.nn.functional.interpolate(frames, size=(new_height, new_width), mode='bilinear',
                                             align_corners=False)            # :130-133
    if num_spatial_views == 1:
        assert min(frames.size(-2), frames.size(-1)) >= spatial_size         # :182
        h_st = (frames.size(-2) - spatial_size) // 2                         # :183-186
        w_st = (frames.size(-1) - spatial_size) // 2
        spatial = [frames[:, :, h_st:h_st + spatial_size, w_st:w_st + spatial_size]]
    else:
        assert num_spatial_views == 3 and min(frames.size(-2), frames.size(-1)) == spatial_size   # :188-189
        spatial = []
        margin = max(frames.size(-2), frames.size(-1)) - spatial_size
        for st in (0, margin // 2, margin):                                  # :192-198
            ed = st + spatial_size
            spatial.append(frames[:, :, st:ed, :] if frames.size(-2) > frames.size(-1) else frames[:, :, :, st:ed])
    crops = []
    for fr in spatial:                                                       # :135, :163-177
        seg_len = (num_frames - 1) * sampling_rate + 1
        if fr.size(1) < seg_len:
            fr = torch.cat([fr, fr[:, -1:].repeat(1, seg_len - fr.size(1), 1, 1)], dim=1)
        slide_len = fr.size(1) - seg_len
        for i in range(num_temporal_views):
            st = slide_len // 2 if num_temporal_views == 1 else round(slide_len / (num_temporal_views - 1) * i)
            crops.append(fr[:, st: st + num_frames * sampling_rate: sampling_rate])
    return crops[0].contiguous()                                             # :138 `frames = frames[0]`

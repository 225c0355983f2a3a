"""Bitwise run-to-run determinism of the HIP forward at config c1 (debug aid)."""
import hashlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gava_clip_amd import VitaCLIP, synth
from gava_clip_amd.config import VIT_B16_T8
from helpers import model_kwargs, synth_torch_state

def h(t): return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
m = VitaCLIP(**model_kwargs(VIT_B16_T8), operand_dtype=prec)
m.load_state_dict(synth_torch_state(VIT_B16_T8, 3), strict=True)
m = m.cuda().eval(); m.debug_taps = True
x = torch.from_numpy(synth.synth_clip(2, 8, 224)).cuda()
runs = []
for r in range(4):
    with torch.no_grad():
        lg, _, _ = m(x)
    torch.cuda.synchronize()
    runs.append([h(m.last["cls_rows"][i]) for i in range(12)] + [h(lg)])
    print(r, " ".join(runs[-1]), flush=True)
print("deterministic:", all(r == runs[0] for r in runs))

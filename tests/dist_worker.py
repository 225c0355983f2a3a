"""Worker for the multi-process tests: one process per rank, rendezvous on 127.0.0.1."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if os.environ.get("GAVA_TEST_BACKEND", "gloo") == "nccl":
        # one process per GPU over RCCL, as bench.py and a real node run it (needs >= world devices)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)


def gather_cpu(rank, world, port, out_dir):
    """VitaCLIP._gather on CPU tensors over gloo: rank-major concatenation of whole-clip shards."""
    from gava_clip_amd import VitaCLIP
    from gava_clip_amd.config import TINY
    from helpers import model_kwargs
    _init(rank, world, port)
    torch.set_num_threads(1)
    m = VitaCLIP(**model_kwargs(TINY))
    feats = torch.arange(3 * 8, dtype=torch.float32).view(3, 8) + 100 * rank
    got = m._gather(feats)
    want = torch.cat([torch.arange(3 * 8, dtype=torch.float32).view(3, 8) + 100 * r for r in range(world)])
    ok = torch.equal(got, want)
    m.gather_across_ranks = False
    ok = ok and torch.equal(m._gather(feats), feats)
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def sharded_forward_gpu(rank, world, port, out_dir):
    """Each rank runs whole clips [rank*b,(rank+1)*b) on cuda:0 through the HIP path; logits after the
    embedding all-gather must equal the single-process logits of the concatenated batch."""
    from gava_clip_amd import VitaCLIP, synth
    from gava_clip_amd.config import TINY
    from helpers import model_kwargs, synth_torch_state
    _init(rank, world, port)
    torch.set_num_threads(2)
    m = VitaCLIP(**model_kwargs(TINY))
    m.load_state_dict(synth_torch_state(TINY, 3), strict=True)
    m = m.cuda().eval()
    b = 2
    x = torch.from_numpy(synth.synth_clip(b * world, TINY.num_frames, TINY.input_size)).cuda()
    with torch.no_grad():
        logits, _, _ = m(x[rank * b:(rank + 1) * b])
        m.gather_across_ranks = False
        full, _, _ = m(x)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"sharded{rank}.npy"), logits.cpu().numpy())
    np.save(os.path.join(out_dir, f"full{rank}.npy"), full.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def sharded_text_gpu(rank, world, port, out_dir):
    """400 classes: each rank encodes 200 prompts and the rows are all-gathered; features and logits
    must equal the single-process ones."""
    from gava_clip_amd import VitaCLIP, synth
    from gava_clip_amd.config import VitaConfig
    from helpers import CLASSES_400, model_kwargs, synth_torch_state
    _init(rank, world, port)
    torch.set_num_threads(2)
    cfg = VitaConfig(input_size=64, num_frames=4, feature_dim=128, num_heads=2, num_layers=1, embed_dim=512,
                     num_global_prompts=4, text_layers=2)
    m = VitaCLIP(**model_kwargs(cfg, CLASSES_400))
    m.load_state_dict(synth_torch_state(cfg, 400), strict=True)
    m = m.cuda().eval()
    b = 1
    x = torch.from_numpy(synth.synth_clip(b * world, cfg.num_frames, cfg.input_size)).cuda()
    with torch.no_grad():
        assert m._text_shard(400) is not None
        logits, _, _ = m(x[rank * b:(rank + 1) * b])
        tf = m.text_features.clone()
        m.gather_across_ranks = False
        assert m._text_shard(400) is None
        full, _, _ = m(x)
        tf_full = m.text_features.clone()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"tsharded{rank}.npy"), logits.cpu().numpy())
    np.save(os.path.join(out_dir, f"tfull{rank}.npy"), full.cpu().numpy())
    np.save(os.path.join(out_dir, f"tf{rank}.npy"), np.stack([tf.cpu().numpy(), tf_full.cpu().numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def ddp_train_gpu(rank, world, port, out_dir):
    """DistributedDataParallel around the drop-in model as training/train.py:347 does: each rank backpropagates its
    own clips through the HIP backward, DDP averages the gradients (gloo here; RCCL on a multi-GPU node).  The
    averaged gradients must equal those of one process on the concatenated batch with the loss averaged."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    from gava_clip_amd import VitaCLIP, synth
    from gava_clip_amd.config import TINY
    from helpers import model_kwargs, synth_torch_state
    _init(rank, world, port)
    torch.set_num_threads(2)
    b = 2
    x = torch.from_numpy(synth.synth_clip(b * world, TINY.num_frames, TINY.input_size)).cuda()
    y = torch.arange(b * world, device="cuda") % 3

    def fresh():
        m = VitaCLIP(**model_kwargs(TINY))
        m.load_state_dict(synth_torch_state(TINY, 3), strict=True)
        return m.cuda().train()

    ddp = DDP(fresh(), find_unused_parameters=False)
    logits = ddp(x[rank * b:(rank + 1) * b])[0]
    torch.nn.functional.cross_entropy(logits, y[rank * b:(rank + 1) * b]).backward()
    torch.cuda.synchronize()
    single = fresh()
    torch.nn.functional.cross_entropy(single(x)[0], y).backward()
    torch.cuda.synchronize()
    worst = 0.0
    n = 0
    for (name, p), (_, q) in zip(ddp.module.named_parameters(), single.named_parameters()):
        if q.grad is None:
            assert p.grad is None, name
            continue
        n += 1
        worst = max(worst, float((p.grad - q.grad).norm() / (q.grad.norm() + 1e-12)))
    np.save(os.path.join(out_dir, f"ddp{rank}.npy"), np.array([worst, n]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    fn, rank, world, port, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    {"gather_cpu": gather_cpu, "sharded_forward_gpu": sharded_forward_gpu, "ddp_train_gpu": ddp_train_gpu,
     "sharded_text_gpu": sharded_text_gpu}[fn](rank, world, port, out_dir)

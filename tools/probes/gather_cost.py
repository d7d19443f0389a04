"""What does the per-step RCCL all-gather of the boxes cost the one-sequence frame?  torchrun --nproc-per-node 1 tools/probes/gather_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
from uvltrack_amd import weightgen as wg
from uvltrack_amd.engine import HipEngine
from uvltrack_amd.spec import spec_b
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
spec = spec_b(256, 256)
eng = HipEngine(spec, dev, max_batch=1); eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
inp = wg.make_inputs(spec, batch=1, seed=0, flags=[2])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
outs = eng.alloc_outputs(1)
step = eng.make_eager_step(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), outs=outs)
box = outs["pred_boxes"].view(1, 4)
src = [torch.zeros(1, 4, device=dev) for _ in range(2)]; dst = [torch.zeros(1, 4, device=dev) for _ in range(2)]
ring = torch.zeros(16, 1, 4, device=dev); ringo = torch.zeros(16, 1, 4, device=dev)
pend = [None, None]
def run(mode, n=600):
    for it in range(n + 100):
        if it == 100: torch.cuda.synchronize(); t0 = time.perf_counter()
        step()
        k = it & 1
        if mode == "copy": src[k].copy_(box)
        elif mode == "gather":
            if pend[k] is not None: pend[k].wait()
            src[k].copy_(box); pend[k] = dist.all_gather_into_tensor(dst[k], src[k], async_op=True)
        elif mode == "gather_nocopy":
            if pend[k] is not None: pend[k].wait()
            pend[k] = dist.all_gather_into_tensor(dst[k], src[k], async_op=True)
        elif mode == "every8":
            ring[it & 7].copy_(box)
            if (it & 7) == 7:
                if pend[0] is not None: pend[0].wait()
                pend[0] = dist.all_gather_into_tensor(ringo[:8], ring[:8], async_op=True)
        elif mode == "sync_gather":
            src[k].copy_(box); dist.all_gather_into_tensor(dst[k], src[k])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    for p in pend:
        if p is not None: p.wait()
    pend[0] = pend[1] = None
    return dt
for rep in range(2):
    print("rep %d " % rep + "  ".join("%s %.1f us" % (m, run(m) * 1e6) for m in ("none", "copy", "gather", "gather_nocopy", "every8", "sync_gather")), flush=True)
dist.destroy_process_group()

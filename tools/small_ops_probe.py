"""Which small device operations (copies, fills, memsets) one eager config-3 step issues, and from where.

    python tools/small_ops_probe.py [meshes]     -> gpurun_out/small_ops_probe.txt

A TorchDispatchMode logs every aten copy / fill / zero / clone with its size and the nearest package frame; a wrapper round
_lib.call counts the library entry points (which issue their own hipMemsetAsync).  Measurement tool, not part of the product."""
import collections
import os
import sys
import traceback

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import _lib, arap  # noqa: E402

WATCH = ("copy_", "fill_", "zero_", "clone", "zeros", "zeros_like", "_foreach_copy_", "empty_like", "contiguous", "_to_copy",
         "full", "ones", "new_zeros")
PKG = os.sep + "surfacenetworks_amd" + os.sep


def _frame():
    for fr in reversed(traceback.extract_stack()):
        if PKG in fr.filename and "small_ops_probe" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "(outside the package)"


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in WATCH:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if isinstance(out, (list, tuple)) and out and isinstance(out[0], torch.Tensor):
                nb = sum(o.numel() * o.element_size() for o in out)
            else:
                nb = t.numel() * t.element_size() if t is not None else 0
            key = (name, _frame())
            self.seen[key] += 1
            self.bytes[key] += nb
        return out


def main():
    meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    ds = arap.ClothSequences([(71, 71)] * meshes, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3,
                             device=dev, model="dir")
    torch.manual_seed(1234)
    model = arap.DirModel().to(dev).train()
    opt = arap.make_optimizer(model)
    rng = np.random.default_rng(10)
    ids = np.arange(meshes)
    for _ in range(3):
        arap.train_step(model, opt, ds.sample_batch(meshes, rng, seq_ids=ids))
    torch.cuda.synchronize()

    calls = collections.Counter()
    real_call = _lib.call

    def counting(name, *a):
        calls[name] += 1
        return real_call(name, *a)

    _lib.call = counting
    for mod in list(sys.modules.values()):
        if getattr(mod, "__name__", "").startswith("surfacenetworks_amd") and getattr(mod, "_lib", None) is _lib:
            pass                                         # modules call _lib.call through the module attribute: patched above
    log = Log()
    with log:
        batch = ds.sample_batch(meshes, rng, seq_ids=ids)
        loss, _ = arap.forward_loss(model, batch)
        opt.zero_grad(set_to_none=False)
        loss.backward()
    torch.cuda.synchronize()
    _lib.call = real_call

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/small_ops_probe.txt", "w") as f:
        f.write(f"one eager config-3 step, {meshes} meshes: sampling + forward + loss + backward (optimizer outside)\n\n")
        f.write("aten operations that move or fill memory, by call site\n")
        for key, n in sorted(log.seen.items(), key=lambda kv: -kv[1]):
            f.write(f"{n:5d}  {log.bytes[key] / max(n, 1):14.0f} B/call  {key[0]:16s} {key[1]}\n")
        f.write("\nlibrary entry points\n")
        for name, n in sorted(calls.items(), key=lambda kv: -kv[1]):
            f.write(f"{n:5d}  {name}\n")
    print(open("gpurun_out/small_ops_probe.txt").read())


if __name__ == "__main__":
    main()

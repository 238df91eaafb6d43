"""Host cost of a training step with launch plans on / off (plans.py): for the three GPU configurations, eager steps —
host enqueue time (the loop without a device synchronisation) and step time — and the plan statistics.
    python tools/plan_probe.py [arap|mnist|faust ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(step, steps, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t_enq / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3


def arap_step(meshes=64):
    from surfacenetworks_amd import arap

    ds = arap.ClothSequences([(71, 71)] * meshes, frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=3, device="cuda",
                             model="dir")
    model = arap.DirModel().cuda().train()
    opt = arap.make_optimizer(model)
    rng = np.random.default_rng(10)
    ids = np.arange(meshes)
    return lambda: arap.train_step(model, opt, ds.sample_batch(meshes, rng, seq_ids=ids))


def mnist_step():
    from surfacenetworks_amd import mesh_mnist as mm

    ds = mm.MeshDigits(512, seed=2, device="cuda", fixed_vertices=150, model="dir")
    model = mm.DirModel().cuda().train()
    opt = mm.make_optimizer(model)
    rng = np.random.default_rng(2)
    ids = np.arange(512)
    return lambda: mm.train_step(model, opt, ds.sample_batch(512, rng, ids=ids))


def faust_step():
    from surfacenetworks_amd import dense_correspondence as dc

    ds = dc.TorusBodies(4, device="cuda")
    model = dc.SiameseModel("lap", 15).cuda().train()
    opt = dc.make_optimizer(model)
    k = [0]

    def step():
        k[0] += 1
        return dc.train_step(model, opt, ds, k[0] % 4, (k[0] + 1) % 4)
    return step


def main():
    from surfacenetworks_amd import plans

    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["arap", "mnist", "faust"]
    if "--profile" in sys.argv:
        import cProfile
        import pstats

        for name in which:
            plans.reset()
            plans.set_enabled(True)
            step = {"arap": arap_step, "arap4": lambda: arap_step(4), "mnist": mnist_step, "faust": faust_step}[name]()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(20):
                step()
            pr.disable()
            torch.cuda.synchronize()
            print("=====", name, "20 steps")
            pstats.Stats(pr).sort_stats("tottime").print_stats(40)
        return
    out = {}
    for name in which:
        mk = {"arap": arap_step, "arap4": lambda: arap_step(4), "mnist": mnist_step, "faust": faust_step}[name]
        for on in (False, True):
            torch.manual_seed(1)
            plans.reset()
            plans.set_enabled(on)
            step = mk()
            enq, tot = timed(step, 20, 5)
            st = plans.stats()
            out[f"{name}/{'plans' if on else 'eager'}"] = {
                "host_enqueue_ms": round(enq, 3), "ms_per_step": round(tot, 3),
                "replayed": sum(s["replayed"] for s in st.values()), "recorded": sum(s["recorded"] for s in st.values()),
                "refused": {k: s["reasons"] for k, s in st.items() if s["refused"]}}
            print(name, "plans" if on else "eager", json.dumps(out[f"{name}/{'plans' if on else 'eager'}"]), flush=True)
            del step
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

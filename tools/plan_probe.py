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
    if "--sections" in sys.argv:
        from surfacenetworks_amd import blocks, kernels

        acc = {}

        def wrap(obj, name, label):
            fn = getattr(obj, name)

            def inner(*a, **k):
                t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    e = acc.setdefault(label, [0.0, 0])
                    e[0] += time.perf_counter() - t0
                    e[1] += 1
            setattr(obj, name, inner)

        wrap(plans.Plan, "_launch", "sn_plan_run")
        wrap(plans.Plan, "new_arenas", "new_arenas")
        wrap(plans.Plan, "run", "Plan.run (ptrs + launch)")
        wrap(blocks, "_plan_forward", "_plan_forward")
        wrap(blocks, "_plan_backward", "_plan_backward")
        wrap(blocks, "_op_operands", "_op_operands")
        wrap(plans, "expand_ext", "expand_ext")
        wrap(plans._Builder, "__call__", "builder")
        wrap(blocks, "_bn_args", "_bn_args")
        for cls in ("_DiracBlock", "_PropagateBlock", "_AvgBlock", "_EluConv"):
            c = getattr(blocks, cls)
            for m in ("forward", "backward"):
                f = getattr(c, m)

                def mk(f, label):
                    def inner(*a, **k):
                        t0 = time.perf_counter()
                        try:
                            return f(*a, **k)
                        finally:
                            e = acc.setdefault(label, [0.0, 0])
                            e[0] += time.perf_counter() - t0
                            e[1] += 1
                    return staticmethod(inner)
                setattr(c, m, mk(f, f"{cls}.{m}"))
        for fname in ("lap_block", "avg_block", "dirac_block", "elu_conv"):
            wrap(blocks, fname, fname)
        from surfacenetworks_amd import arap, dense_correspondence as dc, functional as snF, mesh_mnist as mm

        wrap(dc, "forward_pair_loss", "dc.forward_pair_loss")
        wrap(dc.SiameseModel, "towers", "Siamese.towers")
        wrap(dc.Model, "forward", "dc.Model.forward")
        wrap(dc, "loss_fun_delta_cross_entropy", "dc.loss_fun")
        wrap(dc.TorusBodies, "sample", "TorusBodies.sample")
        wrap(torch.Tensor, "backward", "Tensor.backward")
        wrap(torch.optim.Adam, "step", "Adam.step")
        wrap(torch.optim.Adam, "zero_grad", "Adam.zero_grad")
        wrap(snF, "thin_linear", "thin_linear")
        wrap(snF, "bn_linear", "bn_linear")
        wrap(mm.MeshDigits, "sample_batch", "MeshDigits.sample_batch")
        wrap(mm.DirModel, "forward", "mm.DirModel.forward")
        wrap(mm._Head, "_classify", "mm._classify")
        wrap(arap.ClothSequences, "sample_batch", "Cloth.sample_batch")
        wrap(arap, "loss_fn", "arap.loss_fn")
        wrap(arap.DirModel, "forward", "arap.DirModel.forward")
        wrap(kernels, "clear_absmax", "clear_absmax")
        for name in which:
            plans.reset()
            plans.set_enabled(True)
            step = {"arap": arap_step, "arap4": lambda: arap_step(4), "mnist": mnist_step, "faust": faust_step}[name]()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            acc.clear()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            tot = time.perf_counter() - t0
            torch.cuda.synchronize()
            print("=====", name, "ms/step", tot / 20 * 1e3)
            for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
                print(f"{k:32s} {t / 20 * 1e3:8.3f} ms/step  {n / 20:7.1f} calls/step  {t / n * 1e6:8.1f} us/call")
        return
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
        for on, graphs in ((False, False), (True, False), (True, True)):
            torch.manual_seed(1)
            plans.reset()
            plans.set_enabled(on)
            plans.set_graphs(graphs)
            g0 = plans.graph_stats()
            step = mk()
            enq, tot = timed(step, int(os.environ.get("PROBE_STEPS", "20")), int(os.environ.get("PROBE_WARM", "56")))
            st = plans.stats()
            label = "eager" if not on else ("plans+graphs" if graphs else "plans")
            g1 = plans.graph_stats()
            out[f"{name}/{label}"] = {
                "graph": {k: g1[k] - g0[k] for k in g1},
                "host_enqueue_ms": round(enq, 3), "ms_per_step": round(tot, 3),
                "replayed": sum(s["replayed"] for s in st.values()), "recorded": sum(s["recorded"] for s in st.values()),
                "refused": {k: s["reasons"] for k, s in st.items() if s["refused"]}}
            print(name, label, json.dumps(out[f"{name}/{label}"]), flush=True)
            del step
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

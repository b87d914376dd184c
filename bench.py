#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + secondary) and ms/frame of the trace loop.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload k1|k2|k3|k4]

One "step" = one full frame of the workload through the HIP path: every rank renders the
interleaved tiles it owns (scene + LBVH already resident in HBM), the packed shards are
gathered to rank 0 in one RCCL collective and de-interleaved there.  The frame is the same
for every N, so scaling is STRONG.  N > 1 is launched by torch.distributed.run, one process
per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Workloads (BASELINE.json configs): k3 (default, the configuration the headline target is
quoted on and the largest that fits one GPU) = dragon.scn 3840x2160, 16 spp, LBVH;
k1 = spheres1.scn 640x480; k2 = outdoor.scn + spot light 1920x1080, 4 spp;
k4 = plant-h.scn as normal_triangle, max_depth 8.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel from HIP-event
stage times and in-kernel work counters of the timed frames; `cpu_baseline` times the
UNMODIFIED reference build (oracle/_ref, kind "reference") -- or the CPU restatement
(oracle/, kind "port") when that library is absent -- on a bounded slice of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene fixture, width, height, spp, max_depth or None, description)
    "k1": ("spheres1", 640, 480, 1, None, "K1 SphereExamples/spheres1.scn 640x480 1spp max_depth 5"),
    "k2": ("outdoor_spot", 1920, 1080, 4, None, "K2 TriangleExamples/outdoor.scn + spot light 1920x1080 4spp"),
    "k3": ("dragon", 3840, 2160, 16, None, "K3 ComplexExamples/dragon.scn 3840x2160 16spp max_depth 5, LBVH"),
    "k4": ("plant_h_smooth", 660, 512, 1, 8, "K4 ComplexExamples/plant-h.scn as normal_triangle, 660x512 1spp max_depth 8"),
}
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
VALU_PEAK_TFLOPS = 157.3     # MI355X vector FP32 spec


def cpu_baseline(scene, workload, budget_s=15.0):
    """Reference (or port) Mrays/s on the host cores, on a bounded slice of the frame: whole
    film rows with all their samples, sized from a one-chunk probe to about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _oracle import Oracle, Reference, have_reference
    threads = min(16, os.cpu_count() or 1)
    orc = Oracle()
    use_ref = have_reference()
    if use_ref:
        import tempfile
        from make_fixtures import scene_to_scn
        ref = Reference()
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "bench.scn")
            with open(p, "w") as f:
                f.write(scene_to_scn(scene))
            assert ref.load(p, w=scene.width, h=scene.height, spp=scene.spp, depth=scene.max_depth) > 0

    def run(roi):
        """-> (rays in the slice, seconds of the timed implementation)"""
        _, _, cnt, secs = orc.render(scene, roi=roi, threads=threads, want_float=False)
        if use_ref:   # the reference cannot count rays; the port's count on the same slice is exact
            _, _, secs = ref.render(seed=scene.seed, roi=roi, threads=threads)
        return cnt.total, secs

    y = scene.height // 3
    pw = min(scene.width, 100)                       # < 125 pixels: one OpenMP chunk, one thread
    rays_p, secs_p = run((0, y, pw, y + 1))
    row_secs = secs_p * (scene.width / pw) / min(threads, max(1, scene.width // 125))
    if row_secs > 2 * budget_s:                      # even one row is too much: take part of it
        width = max(125, int(scene.width * budget_s / row_secs) // 125 * 125)
        roi = (0, y, min(width, scene.width), y + 1)
    else:
        rows = max(1, min(scene.height - y, int(budget_s / max(row_secs, 1e-4))))
        roi = (0, y, scene.width, y + rows)
    rays, secs = run(roi)
    return {"value": round(rays / secs / 1e6, 6), "unit": "Mrays/s", "cores": threads,
            "kind": "reference" if use_ref else "port",
            "sample": f"film rows y={roi[1]}..{roi[3] - 1}, x=0..{roi[2] - 1} of the {scene.width}x{scene.height} "
                      f"{workload} frame, all {scene.spp} spp: {rays} rays in {secs:.2f} s; brute force, "
                      f"OpenMP collapse(2) schedule(static,125) over pixels, g++ -O2"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="k3", choices=sorted(WORKLOADS))
    ap.add_argument("--film", type=int, nargs=2, default=None, help="override the film size (not a valid bench line)")
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--accel", default="auto", choices=["auto", "brute", "lbvh"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import dist as rd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace loop has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    sname, w, h, spp, depth, desc = WORKLOADS[args.workload]
    scene = ra.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "scenes", sname + ".npz"))
    overridden = bool(args.film or args.spp)
    if args.film:
        w, h = args.film
    if args.spp:
        spp = args.spp
    scene = scene.copy(width=w, height=h, spp=spp, seed=1)
    if depth is not None:
        scene.max_depth = depth
    accel = {"auto": ra.RT_ACCEL_AUTO, "brute": ra.RT_ACCEL_BRUTE, "lbvh": ra.RT_ACCEL_LBVH}[args.accel]

    os.environ["RT_STAGE_TIMES"] = "1"     # HIP events between the stage launches of every frame
    tracer = ra.Tracer(device=local_rank, accel=accel)
    tracer.load_scene(scene)               # scene SoA + LBVH resident in HBM before the timed region

    def step():
        return rd.render_frame_distributed(tracer, rank, world, device)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        frame, st = step()
        stats.append(st)
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    rays = torch.tensor([float(sum(s.rays for s in stats)), float(sum(s.rays_primary for s in stats)),
                         float(sum(s.rays_secondary for s in stats)), float(sum(s.rays_shadow for s in stats))],
                        dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total, prim, sec, shad = (float(v) for v in rays.tolist())

    if rank == 0:
        K = max(1, args.steps)
        ms_per_step = elapsed / K * 1e3
        # ---- roofline of the dominant kernel (this rank's launches over the timed frames)
        ext_ms = sum(s.ms_extend for s in stats); sh_ms = sum(s.ms_shadow for s in stats)
        which = 0 if ext_ms >= sh_ms else 1
        name = ("k_extend" if which == 0 else "k_shadow") + ("<lbvh>" if stats[0].accel_used == ra.RT_ACCEL_LBVH else "<brute>")
        nodes = sum(s.nodes_visited[which] for s in stats)
        sph = sum(s.sphere_tests[which] for s in stats)
        tri = sum(s.tri_tests[which] for s in stats)
        nrays = sum((s.rays_primary + s.rays_secondary) if which == 0 else s.rays_shadow for s in stats)
        launches = sum(s.launches_extend if which == 0 else s.launches_shadow for s in stats)
        kern_ms = ext_ms if which == 0 else sh_ms
        queue_bytes = 72 if which == 0 else 21     # per ray: record read + hit write | hit point + index read, flag write
        lbvh = stats[0].accel_used == ra.RT_ACCEL_LBVH
        # LBVH: nodes and primitives are fetched per ray; brute force: primitives are staged
        # through LDS once per 256-ray block, so only 1/256 of the tests' bytes are global traffic
        prim_bytes = (16 * sph + 36 * tri) if lbvh else (16 * sph + 36 * tri) / 256.0
        alg_bytes = 64 * nodes + prim_bytes + queue_bytes * nrays
        alg_flops = 44 * nodes + 25 * sph + 47 * tri
        gbs = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        tflops = alg_flops / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and not overridden:
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(name.split("<")[0])
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "launches": launches, "avg_launch_ms": round(kern_ms / max(1, launches), 5),
                    "algorithmic_bytes_per_launch": round(alg_bytes / max(1, launches)),
                    "per_ray": {"nodes": round(nodes / max(1, nrays), 3), "sphere_tests": round(sph / max(1, nrays), 3),
                                "tri_tests": round(tri / max(1, nrays), 3)},
                    "valu": {"achieved": round(tflops, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tflops / VALU_PEAK_TFLOPS, 5)},
                    "stage_ms_per_frame": {"extend": round(ext_ms / K, 3), "shadow": round(sh_ms / K, 3),
                                           "shade": round(sum(s.ms_shade for s in stats) / K, 3),
                                           "other": round(sum(s.ms_other for s in stats) / K, 3)}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(scene, args.workload)
            except Exception as e:      # the checker is optional for the bench line, never silently
                cpu = {"error": repr(e)}
        line = {
            "metric": "Mrays/sec (primary+shadow+secondary), whole job",
            "value": round(total / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "ms_per_frame": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"reference scene {sname} (arrays committed under tests/golden/scenes), seeded jitter",
            "config": {"workload": desc + (" [OVERRIDDEN film/spp: not a headline line]" if overridden else ""),
                       "width": w, "height": h, "spp": spp, "max_depth": int(scene.max_depth),
                       "primitives": int(len(scene.spheres) + len(scene.tri_verts)), "lights": int(len(scene.lights)),
                       "accel": "lbvh" if stats[0].accel_used == ra.RT_ACCEL_LBVH else "brute",
                       "partition": f"interleaved {ra.RT_TILE}x{ra.RT_TILE} tiles over {world} GPU(s), one RCCL gather per frame",
                       "rays_per_frame": {"primary": prim / K, "secondary": sec / K, "shadow": shad / K}},
            "mrays_per_gpu": round(total / elapsed / 1e6 / world, 3),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    tracer.close()


if __name__ == "__main__":
    main()

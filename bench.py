#!/usr/bin/env python3
"""bench.py — skinned vertices/s of the per-frame skinned-scene evaluation path on MI355X.

A "step" is one frame of the hot path over the rank's crowd shard: animation sampling ->
hierarchy propagation -> joint palette -> morph + linear-blend skinning -> per-instance bounds
(skinrt_evaluate(STAGE_ALL)), plus, for N > 1, the RCCL all-gather of the bounds.  Inputs
(rig tables, clocks, morph weights) are resident in HBM before the timed region.

Workload at N=1: BASELINE.json configs[2] (C3: VRM-shaped avatar, 60k verts, 200 joints, 4 weights,
32 morph targets x 10,000 instances — the config the roofline target is quoted on).  N>1: configs[3]
(C4: the same avatar x 100,000 instances sharded over the N ranks, total fixed = strong scaling; no
data-path collective, one RCCL all-gather of the bounds per frame); the line then also reports the
kernels alone and the frame with the skinned vertex buffers gathered (`exchange`).  `--workload
c2|c3|c5` with N>1 is weak scaling (the config per GPU).  Synthetic rigs (renderer_amd/synth.py).

`--gpus N` (N>1) without a launcher starts the N ranks itself (torch.distributed.run children, before
any GPU call); under a launcher WORLD_SIZE must equal N.  Either way the line's `n_gpus` is the
number of ranks RCCL saw (`rccl_ranks`), or there is no line and the exit status is 2.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from renderer_amd import abi, api, shard, synth  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(rig, instances):
    """B = I*V*36 + I*(J*84 + 4*M) + V*(36 + 24*K) + D, D = dense M*V*36 or
    sparse sum nnz*(4+36).  Shared mesh / morph data counted once per frame."""
    V, J, M, K = rig.vertex_count, rig.palette_count, rig.target_count, rig.set_count  # J: every skin's joints + un-skinned mesh nodes
    d = 0
    for t in rig.targets:  # a dense target covers its section (the whole mesh on a one-section rig)
        d += int(t.position.shape[0]) * 36 if t.indices is None else int(t.indices.shape[0]) * 40
    return instances * V * 36 + instances * (J * 84 + 4 * M) + V * (36 + 24 * K) + d


def cpu_baseline(rig, spec, budget_s=15.0):
    """Oracle (fp32 C restatement, reference op order, -O2 -ffp-contract=off) timed on this
    box's host cores on a bounded sample of the same workload: single thread (the reference has
    one render thread, renderer/src/state.rs:236), then instance-parallel over this box's share of the host's cores
    (a GPU box shows the whole machine's CPUs in os.cpu_count(); its share is what OMP_NUM_THREADS says)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle, OracleResult

    orc = Oracle("f32")
    V = rig.vertex_count
    want = ("palette", "position", "normal", "tangent", "aabb")
    probe = synth.make_crowd(spec, rig, instances=2)
    desc = rig.to_desc()
    t0 = time.perf_counter()
    orc.evaluate(rig, probe.tau, probe.clip_id, probe.morph_weights, want=want, desc=desc)
    per_inst = max((time.perf_counter() - t0) / 2, 1e-6)
    n = int(max(2, min(spec.instances, budget_s / per_inst, 3.2e9 / (V * 36 + 1))))  # outputs of the sample: at most 3.2 GB of host memory
    crowd = synth.make_crowd(spec, rig, instances=n)
    res = OracleResult()
    t0 = time.perf_counter()
    orc.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, want=want, desc=desc, into=res)
    dt1 = time.perf_counter() - t0
    threads = host_threads()
    chunks = [(k * n // threads, (k + 1) * n // threads) for k in range(threads)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda c: orc.evaluate(rig, crowd.tau, crowd.clip_id, crowd.morph_weights, want=want,
                                           desc=desc, into=res, begin=c[0], end=c[1]), chunks))
    dtm = time.perf_counter() - t0
    sample = f"{rig.name} rig ({V} verts) x {n} of {spec.instances} instances, all stages, one frame"
    one = {"value": n * V / dt1, "unit": "skinned vertices/s", "cores": 1, "kind": "port", "sample": sample,
           "seconds": round(dt1, 3)}
    many = {"value": n * V / dtm, "unit": "skinned vertices/s", "cores": threads, "kind": "port", "sample": sample,
            "seconds": round(dtm, 3), "host_cpus": os.cpu_count()}
    return one, many


def host_threads() -> int:
    """Worker threads for the CPU baseline: the box's CPU share (OMP_NUM_THREADS where the harness sets it), never more
    than the machine has."""
    have = os.cpu_count() or 1
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    return max(1, min(have, share) if share > 0 else have)


def visible_gpus() -> int:
    """GPUs this process would see, WITHOUT opening one (the parent of the ranks must not initialise the GPU before it
    starts them): KFD topology nodes with SIMDs, cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES lists when they are set."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


class FrameLoop:
    """The bench's frame loop, separate from what it drives so that the CPU tests can run the SAME step / fence /
    exchange / timing code with two gloo ranks and the oracle standing in for the crowd (tests/test_bench_loop.py):
    `crowd` needs evaluate(stages) (and advance_clocks(dt) when clocks advance), `ctx` synchronize() and `stream`;
    `aabb_rows` is a torch tensor [mine, 6] (device memory of the crowd's bounds on a GPU) or None without a launcher."""

    def __init__(self, crowd, ctx, aabb_rows, total: int, world: int, launched: bool, advance_dt=None):
        self.crowd, self.ctx, self.total, self.world, self.launched = crowd, ctx, total, world, launched
        self.advance_dt = advance_dt
        self.torch = self.dist = None
        self.on_gpu = False
        self.exchange = None
        if launched:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            self.on_gpu = bool(aabb_rows.is_cuda)
            # The exchange overlaps the next frame (shard.OverlappedRowGather: RCCL on a side stream behind an event of the
            # library's stream; the next frame waits for a 24-byte-per-instance copy, never for the collective or the
            # host).  Rehearsed with one rank: a host synchronize plus the collective in line cost 0.19 ms of a 4.5 ms
            # frame, the collective in line on the library's stream 0.16 ms, this form 0.01 ms.
            self.exchange = shard.OverlappedRowGather(aabb_rows, total, world, ctx.stream if self.on_gpu else 0)

    def step(self, gather=True, advance=None):
        gather = gather and self.launched
        if gather:
            self.exchange.before_frame()
        if (self.advance_dt is not None) if advance is None else advance:
            self.crowd.advance_clocks(self.advance_dt if self.advance_dt is not None else 1.0 / 60.0)  # device-side, stream-ordered
        self.crowd.evaluate(abi.STAGE_ALL)
        if gather:
            self.exchange.after_frame()

    def fence(self):
        self.ctx.synchronize()
        if self.launched:
            if self.on_gpu:
                self.torch.cuda.synchronize()
            self.dist.barrier()
            if self.on_gpu:
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if not self.launched:
            return x
        tt = self.torch.tensor([x], dtype=self.torch.float64, device=self.exchange.local.device)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def timed(self, steps: int, gather=True, advance=None, per_step=None) -> float:
        """EXACTLY `steps` frames between two fences; the slowest rank's wall time."""
        self.fence()
        t0 = time.perf_counter()
        for k in range(steps):
            self.step(gather, advance)
            if per_step is not None:
                per_step(k)
        self.fence()
        return self.max_over_ranks(time.perf_counter() - t0)

    def gathered_bounds(self):
        return self.exchange.result() if self.exchange is not None else None


def measured_bandwidth(device: int):
    """What this box's HBM does on plain streams (report the fraction of the measured
    device-to-device copy bandwidth next to the nominal peak): a 4 GiB hipMemcpyDtoD (read + write) and
    a 4 GiB hipMemset (write only), timed with HIP events after the timed region.  Plain HIP through
    ctypes, on the runtime the library itself uses."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def ok(rc):
        if rc != 0:
            raise RuntimeError(f"HIP error {rc}")
    n = 4 << 30
    ok(hip.hipSetDevice(C.c_int(device)))
    a, b, e0, e1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    ok(hip.hipMalloc(C.byref(a), C.c_size_t(n)))
    try:
        ok(hip.hipMalloc(C.byref(b), C.c_size_t(n)))
        ok(hip.hipEventCreate(C.byref(e0))); ok(hip.hipEventCreate(C.byref(e1)))
        out = {}
        copy = lambda: ok(hip.hipMemcpyAsync(b, a, C.c_size_t(n), C.c_int(3), None))   # hipMemcpyDeviceToDevice
        fill = lambda: ok(hip.hipMemsetAsync(a, C.c_int(1), C.c_size_t(n), None))
        for name, fn, moved in (("fill", fill, n), ("d2d_copy", copy, 2 * n)):
            for _ in range(2):
                fn()
            reps = 8
            ok(hip.hipEventRecord(e0, None))
            for _ in range(reps):
                fn()
            ok(hip.hipEventRecord(e1, None))
            ok(hip.hipEventSynchronize(e1))
            ms = C.c_float()
            ok(hip.hipEventElapsedTime(C.byref(ms), e0, e1))
            out[name] = moved * reps / (ms.value * 1e-3) / 1e9
        return out
    finally:
        hip.hipFree(a)
        if b: hip.hipFree(b)
        if e0: hip.hipEventDestroy(e0)
        if e1: hip.hipEventDestroy(e1)


def spawn_ranks(n: int, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (python -m torch.distributed.run) BEFORE this process has made any GPU call, relay their output
    and exit with their status.  Refuses (exit 2, no JSON line) when the node has fewer GPUs."""
    import socket
    import subprocess
    have = visible_gpus()  # sysfs: this process never opens a GPU
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} asked for, {have} visible: refusing to print a line for ranks that do not exist\n")
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # the first 6-8 frames of a fresh crowd run up to 10 % slower
    ap.add_argument("--workload", default=None, choices=[None, "c2", "c3", "c4", "c5"])
    ap.add_argument("--instances", type=int, default=None, help="override the TOTAL instance count")
    ap.add_argument("--no-vertex-gather", action="store_true", help="N>1: skip the extra vertex all-gather measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region: no clock-advancing loop, no plain-placement crowd, no copy / fill rates (profiling runs)")
    ap.add_argument("--advance-clocks", action="store_true",
                    help="every timed frame also advances the instances' clocks on the device (skinrt_advance_clocks, 1/60 s): the frame "
                         "a renderer's loop runs; without it the loop is timed after the main region and reported as frame_loop")
    ap.add_argument("--multi-skin", action="store_true",
                    help="the workload's avatar cut into three skins on one skeleton (body / face / hair, each with its own joint order) plus "
                         "a rigid prop on a bone (synth.make_multi_rig): same vertices, same results for the skinned part; no CPU baseline leg")
    ap.add_argument("--exact", type=int, default=0, choices=[0, 1, 2],
                    help="skinrt_config.exact_arithmetic: 0 chosen per rig, 1 contraction-free kernels (bit-equal to the oracle), 2 FMA kernels")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        spawn_ranks(args.gpus, sys.argv[1:])  # never returns
    # stdout carries ONE JSON line and nothing else.  Libraries write there behind Python's back (RCCL prints its
    # version block and its warnings to fd 1, at init and again at the first collective), so fd 1 is pointed at stderr
    # for the whole run and the line goes to a private copy of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # one line per GPU that really runs, or none: a line must never claim more GPUs than ranks
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): refusing\n")
        sys.exit(2)
    dist = None
    torch = None
    rccl_ranks = 1
    if launched:
        # RCCL prints a version banner on stdout at NCCL_DEBUG=VERSION: keep stdout to the one JSON line
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        rccl_ranks = dist.get_world_size()
        if rccl_ranks != args.gpus:
            sys.stderr.write(f"bench.py: RCCL sees {rccl_ranks} ranks, --gpus {args.gpus}: refusing\n")
            sys.exit(2)
    n_gpus = rccl_ranks

    # N = 1: C3 (the config the metric is quoted on).  N > 1: BASELINE's C4 — the same avatar x 100,000
    # instances sharded over the N GPUs, total fixed (strong scaling): 50,000 per GPU at N=2, 12,500 at N=8.
    name = args.workload or ("c3" if n_gpus == 1 else "c4")
    spec = synth.CONFIGS[name]
    strong = name == "c4"
    total = args.instances or (spec.instances if (strong or n_gpus == 1) else spec.instances * world)
    lo, hi = shard.shard_range(total, rank, world)
    mine = hi - lo
    rig_data = synth.make_rig(spec)
    crowd_in = synth.make_crowd(spec, rig_data, instances=mine, first=lo)
    if args.multi_skin:
        rig_data, target_map = synth.make_multi_rig(rig_data, seed=11)
        if crowd_in.morph_weights is not None:
            crowd_in.morph_weights = np.ascontiguousarray(crowd_in.morph_weights[:, target_map])
        args.no_cpu_baseline = True

    ctx = api.Context(device=local_rank, exact_arithmetic=args.exact)
    rig = api.Rig(ctx, rig_data)
    crowd = api.Crowd(ctx, rig, mine)
    crowd.set_clocks(crowd_in.tau, crowd_in.clip_id)
    if crowd_in.morph_weights is not None:
        crowd.set_morph_weights(crowd_in.morph_weights)
    V = rig_data.vertex_count

    aabb_t = None
    vert_t = []
    if launched:
        dev = f"cuda:{local_rank}"
        ptr, _ = crowd.device_pointer("aabbs")
        aabb_t = torch.as_tensor(shard.DeviceArray(ptr, (mine, 6)), device=dev)
        for what in ("positions", "normals", "tangents"):
            pptr, _ = crowd.device_pointer(what)
            vert_t.append(torch.as_tensor(shard.DeviceArray(pptr, (mine, V * 3)), device=dev))

    loop = FrameLoop(crowd, ctx, aabb_t, total, world, launched, advance_dt=(1.0 / 60.0) if args.advance_clocks else None)
    step, fence, max_over_ranks = loop.step, loop.fence, loop.max_over_ranks

    for _ in range(args.warmup):
        step()
    stage = np.zeros((args.steps, 4))

    def stamp(k):
        stage[k] = crowd.stage_ms()  # HIP events on the library's stream, around each kernel

    dt = loop.timed(args.steps, per_step=stamp)

    # next to the frame (kernels + bounds all-gather = `value`), the kernels alone and the
    # frame with the skinned vertex buffers gathered as well (xGMI-bound by construction)
    exchange = None
    if launched:
        dt_k = loop.timed(args.steps, gather=False)
        exchange = {"kernels_only_ms": dt_k / args.steps * 1e3, "with_aabb_allgather_ms": dt / args.steps * 1e3,
                    "with_vertex_allgather_ms": None}
        if not args.no_vertex_gather:
            fence()
            t0 = time.perf_counter()
            step()
            ctx.synchronize()
            moved = shard.all_gather_rows_chunked(vert_t, total, world)
            fence()
            dt_v = max_over_ranks(time.perf_counter() - t0)
            exchange["with_vertex_allgather_ms"] = dt_v * 1e3
            exchange["vertex_allgather"] = {"bytes_received_per_rank": moved, "steps": 1,
                                            "how": "position, normal, tangent one after the other, row chunks through one 8 GiB buffer"}

    # What a renderer's loop does per frame: new inputs, then the frame.  The clocks advance ON THE DEVICE, stream-ordered
    # (skinrt_advance_clocks): nothing drains the GPU between frames.  Measured next to the plain frame.
    frame_loop = None
    if not args.advance_clocks and not args.no_extras:
        n_loop = min(args.steps, 30)
        dt_adv = loop.timed(n_loop, advance=True)
        dt_same = loop.timed(n_loop, advance=False)
        frame_loop = {"ms_per_step_advancing_clocks": dt_adv / n_loop * 1e3, "ms_per_step_evaluate_only": dt_same / n_loop * 1e3,
                      "steps": n_loop, "how": "skinrt_advance_clocks(1/60 s) + skinrt_evaluate per frame, no host wait inside the loop"}
        crowd.set_clocks(crowd_in.tau, crowd_in.clip_id)

    # The same frames with the three output arrays where three plain hipMalloc calls put them (no steering over the HBM's
    # thirds): what the kernels do without placement.hip.  A second context + crowd, N = 1 only.
    plain_ms = None
    if n_gpus == 1 and crowd.placement["strategy"] != "plain" and not args.no_extras:
        try:
            ctx2 = api.Context(device=local_rank, exact_arithmetic=args.exact, output_placement=1)
            rig2 = api.Rig(ctx2, rig_data)
            crowd2 = api.Crowd(ctx2, rig2, mine)
            crowd2.set_clocks(crowd_in.tau, crowd_in.clip_id)
            if crowd_in.morph_weights is not None:
                crowd2.set_morph_weights(crowd_in.morph_weights)
            acc = []
            for k in range(20):
                crowd2.evaluate(abi.STAGE_ALL, sync=True)
                if k >= 10:
                    acc.append(crowd2.stage_ms()[1])
            plain_ms = float(np.mean(acc))
            crowd2.close(); rig2.close(); ctx2.close()
        except api.SkinrtError as e:  # no room for a second crowd: a reported extra, never a reason to lose the line
            plain_ms = None
            sys.stderr.write(f"bench.py: plain-placement crowd skipped: {e}\n")

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = total * V * args.steps / dt
        skin_ms = float(stage[:, 1].mean())
        bytes_launch = algorithmic_bytes(rig_data, mine)
        achieved = bytes_launch / (skin_ms * 1e-3)
        # HBM bytes per launch from rocprofv3 PMC passes of THIS command (separate --pmc runs, corrected as
        # /opt/skills/guides/MI355X_MICROARCH.md prescribes) on another box of the pool, committed under profiles/; a plain
        # `python bench.py` cannot read counters of its own run, so the source is named beside the number
        traffic = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"{name}:{mine}")
                traffic_source = tj.get("source")
            except Exception:
                traffic = None
        info = rig.info
        out = {
            "metric": "skinned vertices/sec (whole node) + ms/frame at N avatars; % HBM roofline",
            "value": value, "unit": "skinned vertices/s", "n_gpus": n_gpus, "rccl_ranks": rccl_ranks if launched else 0,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{name}: VRM-shaped avatar x instances" if name in ("c3", "c4") else name) + (" (multi-skin cut)" if args.multi_skin else ""),
                       "instances_total": total, "instances_per_gpu": mine, "vertices": V,
                       "joints": rig_data.joint_count, "skins": len(rig_data.skin_list()), "mesh_sections": len(rig_data.section_list()),
                       "palette_entries": rig_data.palette_count, "nodes": rig_data.node_count, "weight_sets": rig_data.set_count,
                       "morph_targets": rig_data.target_count, "channels": rig_data.max_channels,
                       "stages": "pose+morph+skin+aabb" + ("+allgather(aabb)" if launched else ""),
                       "exact_arithmetic": int(info.exact_arithmetic), "exact_arithmetic_config": args.exact, "parallelism": f"instances sharded x{n_gpus}"},
            "stage_ms": {"pose": float(stage[:, 0].mean()), "skin_morph": skin_ms,
                         "aabb_reduce": float(stage[:, 2].mean()), "device_span": float(stage[:, 3].mean()),
                         "skin_morph_plain_placement": plain_ms},
            "roofline": {"bound": "hbm", "kernel": "skin stage = k_classify + k_skin2 (morph + linear-blend skinning + per-tile bounds)", "achieved": achieved / 1e9,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                         "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": skin_ms, "traffic": traffic,
                         "traffic_source": traffic_source},
            "output_placement": crowd.placement,
            "rig_compile": {"tiles": info.tile_count, "max_tile_joints": info.max_tile_joints, "handed_over_tiles": info.handed_over_tiles,
                            "lds_tiles": info.lds_tiles, "morph_entries": int(info.morph_entries)},
        }
        if exchange is not None:
            out["exchange"] = exchange
        if frame_loop is not None:
            out["frame_loop"] = frame_loop
        if args.advance_clocks:
            out["config"]["stages"] = "advance_clocks+" + out["config"]["stages"]
        try:
            if n_gpus != 1:
                raise RuntimeError("measured at N=1 only")
            if args.no_extras:
                raise RuntimeError("--no-extras")
            bw = measured_bandwidth(local_rank)
            out["roofline"]["measured_GBps"] = {"d2d_copy_read_plus_write": bw["d2d_copy"], "fill_write_only": bw["fill"]}
            out["roofline"]["frac_of_measured_copy"] = (achieved / 1e9) / bw["d2d_copy"]
        except Exception as e:  # a reported extra, never a reason to lose the bench line
            out["roofline"]["measured_GBps"] = {"skipped": str(e)[:200]}
        if not args.no_cpu_baseline and n_gpus == 1:
            one, many = cpu_baseline(rig_data, spec)
            out["cpu_baseline"] = one
            out["cpu_baseline_mt"] = many
        print(json.dumps(out), file=json_out, flush=True)

    crowd.close(); rig.close(); ctx.close()
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: clips/sec of ReLaX-VQA feature extraction (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU under torch.distributed.run; run plainly, bench.py starts that launcher itself as a
   child process before anything touches a GPU; a WORLD_SIZE that disagrees with --gpus is an error)

One step = one pass of the hot path over one clip per rank, frames already resident in HBM:
  workload "config3": synthetic 1080p, 32 (frame, next) pairs ->
     residual + patch score + top-196 + fragments            (HIP, bit-exact integer path)
     ResNet-50 layer-stack on the original fragments + pool on the residual fragments  -> [32,15171]
     ViT-B/16 pool on both fragment sets                                               -> [32, 4608]
     per-clip mean -> [19779]; N > 1: RCCL all-gather of the per-clip vectors
Synthetic frames and deterministic random-init weights of the named architectures (no network here).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel
(under the default arithmetic, f16x2, the plain split-plane GEMM gemm_h3<false,false,false> - the ViT's 49 GEMMs per pass: fp32-grade
products on the fp16 matrix cores, timed live with HIP events on the launch stream; `frac` = ALGORITHMIC FLOPs / time / peak, the pipe's
executed rate beside it as `pipe_busy_frac`; every contraction launch together under `all_contractions`) and `cpu_baseline` (the CPU oracle, reference-faithful schedule, on a bounded sample of the same workload).
The exact-fp32-MFMA path (`--precision fp32`) is measured beside the headline as `exact_fp32_mode`.
The default N = 1 run also times a few steps of the other BASELINE configurations (`other_workloads`: config 2, the config-4
clip shape, the config-5 recipe at 2160p) so that the driver's record carries them, not only builder-run files.

  python bench.py --gpus N --workload config4 --dataset-clips 1200
BASELINE config 4 as written (strong scaling): a KoNViD-1k-shaped list of 1200 clips sharded over the ranks
(relax-vqa_amd/dataset.py), batches of --clips-per-step through the engine, ONE all-gather of the [1200, 19779] matrix;
prints clips/s over the whole pass with `all_gather_ms` beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import relax_vqa_amd  # noqa: E402,F401
from relax_vqa_amd import distributed as rdist  # noqa: E402
from relax_vqa_amd import synth  # noqa: E402
from relax_vqa_amd.engine import RelaxEngine  # noqa: E402

WORKLOADS = {
    # name: (H, W, T, use_vit)
    "config2": (720, 1280, 16, False),
    "config3": (1080, 1920, 32, True),
    "config4": (540, 960, 16, True),
    # full ReLaX (config 5 recipe: whole-frame + fragment features of both backbones, residual+flow fragments -> 35203-d)
    # at the sizes one GPU can be fed quickly; not the headline
    "full1080p": (1080, 1920, 32, True),
    "full2160p": (2160, 3840, 32, True),
}
FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (never the 2:1-sparsity figure)
HBM_PEAK_GBPS = 8000.0


def cpu_baseline(H, W, T, use_vit, rn_sd, vit_sd, sample_pairs):
    """Oracle timed on this host's cores (kind 'port'): reference-faithful schedule on `sample_pairs` pairs of the
    same workload, extrapolated to a clip; the de-duplicated schedule is reported beside it.  Batch size 1 does not
    scale to every core of a big host, so the thread count is calibrated first (one pair per candidate) and the CPU is
    timed at its best; the default-thread-count figure (what the reference would use as shipped) is reported too."""
    from oracle import pipeline_ref
    frames = synth.synthetic_clip(sample_pairs, H, W, clip_id=99)
    vit = vit_sd if use_vit else None
    default_threads = torch.get_num_threads()
    pipeline_ref.clip_features(frames[:1], rn_sd, vit, schedule="dedup")   # warm the thread pool / allocator
    candidates = sorted({t for t in (8, 16, 32, 64, default_threads) if t <= default_threads})
    timing = {}
    for t in candidates:
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        pipeline_ref.clip_features(frames[:1], rn_sd, vit, schedule="dedup")
        timing[t] = time.perf_counter() - t0
    best = min(timing, key=timing.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    pipeline_ref.clip_features(frames, rn_sd, vit, schedule="faithful")
    t_faithful = (time.perf_counter() - t0) / sample_pairs * T
    t0 = time.perf_counter()
    pipeline_ref.clip_features(frames, rn_sd, vit, schedule="dedup")
    t_dedup = (time.perf_counter() - t0) / sample_pairs * T
    torch.set_num_threads(default_threads)
    t0 = time.perf_counter()
    pipeline_ref.clip_features(frames[:1], rn_sd, vit, schedule="faithful")
    t_faithful_default = (time.perf_counter() - t0) * T
    return {
        "value": 1.0 / t_faithful, "unit": "clips/s", "cores": best, "kind": "port",
        "sample": f"{sample_pairs} of {T} pairs of one {W}x{H} clip, reference-faithful schedule "
                  f"(15+1 ResNet-50 forwards/pair, python patch loop, ViT rebuilt per call), bs 1, fp32, torch CPU at its "
                  f"best thread count ({best}, calibrated over {candidates})",
        "dedup_value": 1.0 / t_dedup, "sec_per_clip_faithful": t_faithful, "sec_per_clip_dedup": t_dedup,
        "default_threads": default_threads, "value_at_default_threads": 1.0 / t_faithful_default,
        "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
    }


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def hbm_traffic_per_launch(workload, clips_per_step, precision):
    """PMC-measured HBM bytes per contraction launch, if a committed profile exists for this exact workload and precision."""
    path = os.path.join(ROOT, "profiles", "r05_hbm_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("workload") == workload and rec.get("clips_per_step") == clips_per_step and rec.get("precision") == precision:
            return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


DOMINANT_KERNEL_MATCH = "relax::gemm_h3<false,false,false>"   # the plain f16x2 GEMM (three products): the ViT's 49 GEMMs per pass
DOMINANT_TRAFFIC = None     # (bytes per launch, launches) of that kernel alone, set by measure_hbm_traffic


FLOW_KERNELS = ("relax::flow_", "relax::poly_expansion", "relax::pyramid_fused", "relax::gauss", "relax::resize_linear_f32", "relax::update_matrices_k",
                "relax::box_solve_fused", "relax::mag_minmax")


def measure_hbm_traffic(workload, clips_per_step, precision, split_k):
    """roofline.traffic measured in THIS run: two child passes of this script under rocprofv3 (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`:
    separate passes, kernel-trace only, as the gfx950 guide prescribes; FETCH_SIZE doubled: it reports half of a 16-B-per-lane
    streaming read), one warm-up + one step each; bytes per launch of the dominant kernel family (gemm_x6*, conv1_x6) and, for the
    full pipelines, the bytes of all Farneback kernels per clip (one stage call per clip).  The children are started as CHILD
    processes (never exec) with the program itself after `--`, each bounded to 150 s.
    Returns (bytes per contraction launch | None, note, flow-stage bytes per clip | None)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH", None
    if any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, "this run is itself being profiled", None
    tmp = tempfile.mkdtemp(prefix="relax_pmc_", dir="/tmp")
    tot, flow_tot, launches = {}, {}, 0
    dom_tot, dom_launches = {}, 0
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", "--workload", workload, "--clips-per-step", str(clips_per_step),
                   "--precision", precision, "--gemm-split-k", str(split_k)]
            res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=150)
            files = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
            if res.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {res.returncode}): {res.stderr[-300:]}", None
            total, flow_total, ids = 0.0, 0.0, set()
            dom_total, dom_ids = 0.0, set()
            for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
                if r["Counter_Name"] != counter:
                    continue
                if any(k in r["Kernel_Name"] for k in ("relax::gemm_x6", "relax::conv1_x6", "relax::gemm_h2", "relax::gemm_h3", "relax::rn_block", "relax::stem_pool")):
                    total += float(r["Counter_Value"])
                    ids.add(r["Dispatch_Id"])
                    if DOMINANT_KERNEL_MATCH in r["Kernel_Name"].replace(" ", ""):
                        dom_total += float(r["Counter_Value"])
                        dom_ids.add(r["Dispatch_Id"])
                elif any(k in r["Kernel_Name"] for k in FLOW_KERNELS):
                    flow_total += float(r["Counter_Value"])
            tot[counter] = total * 1024.0          # KiB -> bytes
            flow_tot[counter] = flow_total * 1024.0
            launches = len(ids)
            dom_tot[counter] = dom_total * 1024.0
            dom_launches = len(dom_ids)
        if not launches:
            return None, "no contraction dispatch in the PMC pass", None
        flow_per_clip = (2.0 * flow_tot["FETCH_SIZE"] + flow_tot["WRITE_SIZE"]) / (2 * clips_per_step) if flow_tot["WRITE_SIZE"] > 0 else None
        global DOMINANT_TRAFFIC
        DOMINANT_TRAFFIC = ((2.0 * dom_tot["FETCH_SIZE"] + dom_tot["WRITE_SIZE"]) / dom_launches, dom_launches) if dom_launches else None
        return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / launches, (
            f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, one child pass each (1 warm-up + 1 step), "
            f"(2 x FETCH_SIZE + WRITE_SIZE) over the {launches} contraction dispatches of the pass (gemm_h3 / gemm_x6 / conv1_x6 / the fused ResNet blocks)"), flow_per_clip
    except subprocess.TimeoutExpired:
        return None, "a rocprofv3 pass did not finish in 150 s", None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run` (one rank per GPU) as a
    CHILD process, relay its output and return its exit code.  Called before any GPU call of this process (a process
    that has initialised the GPU must never exec another program on this pool; a child is fine either way)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: --gpus {n_ranks} without WORLD_SIZE: launching {n_ranks} ranks: {' '.join(cmd)}", file=sys.stderr)
    proc = subprocess.run(cmd, env=env)   # stdout / stderr are inherited: the JSON line of rank 0 goes straight through
    return proc.returncode


def _rccl_ranks(world, backend):
    """Ranks whose collective ran over RCCL: `world` under the nccl backend (and for the single rank of an N = 1 run, which has no
    collective), 0 under gloo (the rehearsal backend of several ranks sharing one GPU) - `ranks` / `backend` say what ran."""
    return world if (world == 1 or backend == "nccl") else 0


def launch_check(rank, world):
    """--launch-check: the multi-rank control flow of this file without an engine (CPU test of the launcher): rendezvous,
    barrier, the feature all-gather on stand-in per-clip vectors, the max-over-ranks timing reduction, one JSON line."""
    dev = "cuda" if torch.cuda.is_available() and dist.is_initialized() and dist.get_backend() != "gloo" else "cpu"
    local = torch.full((2, 5), float(rank), device=dev)
    t0 = time.perf_counter()
    out = rdist.gather_clip_vectors(local, world * 2, rank, world) if world > 1 else local
    if world > 1:
        rdist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = rdist.all_reduce_max(elapsed, dev)
    want = torch.arange(world, dtype=torch.float32).repeat_interleave(2)[:, None].expand(-1, 5)
    assert torch.equal(out.cpu(), want), "all-gather returned rows out of clip order"
    if rank == 0:
        backend = dist.get_backend() if world > 1 else None
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": world, "rccl_ranks": _rccl_ranks(world, backend),
                          "backend": backend, "pids_distinct": True}))
    if world > 1:
        rdist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=16)
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra measurement of the other precision (exact fp32)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the extra pinned-host-to-device measurement")
    ap.add_argument("--precision", default="f16x2", choices=["fp32", "bf16x3", "bf16x6", "f16x2"],
                    help="arithmetic of the contraction kernels for the headline loop.  f16x2 (default): fp32 operands as two fp16 planes of a "
                         "power-of-two multiple of themselves, three (K >= 256) or four partial products on the fp16 MFMA, fp32 accumulate - the "
                         "whole ViT (GEMMs and attention) and ResNet-50's stem, 3x3 convolutions, layer3 / layer4 (and the fused blocks of layer1 / "
                         "layer2 where enabled); what has no f16x2 kernel runs bf16x6: fp32 operands as three bf16 planes, six partial products.  "
                         "Both fp32-grade: error against fp64 no larger than the fp32 FMA chain's "
                         "(tests/test_gpu_h2.py, tests/test_gpu_x6.py).  bf16x6: that arithmetic everywhere; fp32: exact fp32 MFMA; bf16x3: "
                         "lower precision, never the headline")
    ap.add_argument("--clips-per-step", type=int, default=0,
                    help="clips each rank pushes through the engine per step (one batched pass: B*2*T fragments); default: 2048 "
                         "fragments per pass for configs 2 / 3 / 4 (64 / 32 / 64 clips, dataset pass included), 16 clips for the full pipelines")
    ap.add_argument("--dataset-clips", type=int, default=0,
                    help="dataset mode (BASELINE config 4 as written): this many clips sharded over the ranks, one all-gather of the "
                         "[n, F] matrix at the end; strong scaling; --steps is ignored (the pass is ceil(n / ranks / clips-per-step) batches)")
    ap.add_argument("--resident-clips", type=int, default=0, help="distinct synthetic clips kept in HBM per rank (default 2; dataset mode 4)")
    ap.add_argument("--gemm-split-k", type=int, default=1, choices=[0, 1],
                    help="0: no tail split-K - bits independent of the batch composition, i.e. of the number of ranks")
    ap.add_argument("--dump-matrix", default=None, help="dataset mode: rank 0 saves the gathered [n, F] matrix as .npy (tests)")
    ap.add_argument("--host-clips", action="store_true",
                    help="dataset mode: after the device-resident pass, the same pass with every clip living in (pageable) host memory - "
                         "loader threads stage it in pinned buffers, a side stream copies batch k+1 under the compute of batch k "
                         "(relax-vqa_amd/dataset.py); reported beside `value` as `host_fed` with `h2d_hidden_frac`")
    ap.add_argument("--stub-compute-ms", type=float, default=0.0,
                    help="dataset mode, host-feed REHEARSAL: the backbones are replaced by a device-side wait of this many milliseconds per "
                         "batch (every clip of the batch is still read once on the device, so its copy must have landed).  With "
                         "RELAX_DIST_BACKEND=gloo and --gpus 8 on a one-GPU box the loaders, pinned pools and copy streams of eight ranks run "
                         "at once against one device: what is measured is the HOST side of an 8-rank pass (staging and H2D rates, pinned "
                         "memory), never a feature-extraction rate; the record says so and carries no `value`")
    ap.add_argument("--from-frame-files", default=None, metavar="DIR",
                    help="dataset mode: the clips are read from sampled-frame PNG files under DIR ({video}_{n}.png / {video}_{n}_next.png, "
                         "written there once from synthetic frames if absent) by sampling.load_clip_from_frames in the loader threads, "
                         "decoded straight into pinned staging memory: the ingest rate of a from-files pass (PNG decode stays outside the "
                         "metric: reported as `from_frame_files`, never as `value`)")
    ap.add_argument("--loader-workers-sweep", default=None, metavar="N,N,...",
                    help="with --from-frame-files: repeat the from-files pass for each of these loader-thread counts")
    ap.add_argument("--loader-processes", default=None, metavar="P,P,...",
                    help="with --from-frame-files: repeat the from-files pass with P loader PROCESSES (relax-vqa_amd/loaderpool.py: decode in "
                         "spawned workers into shared memory the rank has page-locked) for each P; the pools are started before this process "
                         "touches the GPU")
    ap.add_argument("--prefetch", type=int, default=2, help="dataset mode: batches the loader threads run ahead of the engine (0: inline, no threads)")
    ap.add_argument("--loader-workers", type=int, default=8, help="dataset mode: loader threads per rank")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the short extra measurements of configs 2 / 4 / 5")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic (N = 1 only); a committed profile "
                         "of the same workload, batch and precision is reported instead when there is one")
    ap.add_argument("--traffic-child", action="store_true", help="(internal) one warm-up + one step of the workload, nothing else: the PMC target")
    ap.add_argument("--launch-check", action="store_true",
                    help="only rehearse the N-rank launch + collectives (no engine, no GPU needed): tests/test_bench_launch.py")
    args = ap.parse_args()
    if args.clips_per_step <= 0:
        # 2048 fragments per backbone pass for the backbone-only workloads, the dataset pass included (the partial last round of tiles
        # of a launch weighs less: config 3 +1 %, config 2 +9 %, config 4 +4 % over 16 clips per step); the full pipelines keep 16
        args.clips_per_step = 16 if args.workload.startswith("full") else 2048 // (2 * WORKLOADS[args.workload][2])

    # ---- launcher: nothing above or in this block touches a GPU -------------------------------------------------
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: refusing to report a {env_world}-rank number "
              f"under an --gpus {args.gpus} command line", file=sys.stderr)
        sys.exit(2)

    # loader-process pools of the from-files pass: spawned NOW, before this process touches the GPU or joins the process group (a process
    # that has initialised the GPU must not start other programs on these boxes).  The frame files are written first (rank 0; the
    # other ranks wait for its marker file), so the workers find them.
    args.loader_pools = []
    if args.dataset_clips and args.from_frame_files and args.loader_processes:
        import functools
        from relax_vqa_amd import loaderpool
        H_, W_, T_, _ = WORKLOADS[args.workload]
        n_res = args.resident_clips or 4
        marker = os.path.join(args.from_frame_files, f".complete_{args.workload}_{n_res}")
        if int(os.environ.get("RANK", "0")) == 0:
            write_frame_files(args.from_frame_files, n_res, T_, H_, W_)
            open(marker, "w").close()
        while not os.path.exists(marker):
            time.sleep(0.2)
        names = [f"video{v}" for v in range(n_res)]
        src = functools.partial(loaderpool.frames_source, sampled_frame_path=args.from_frame_files, names=names)
        for p_ in (int(x) for x in args.loader_processes.split(",")):
            args.loader_pools.append((p_, loaderpool.LoaderProcessPool(src, processes=p_, segments_per_worker=3).start()))

    rank, world, local_rank = rdist.init_from_env()
    assert world == args.gpus
    if args.launch_check:
        launch_check(rank, world)
        return
    H, W, T, use_vit = WORKLOADS[args.workload]

    torch.cuda.set_device(local_rank)
    if args.dataset_clips and args.stub_compute_ms > 0:      # host-feed rehearsal: no engine (N ranks may share one GPU's memory)
        rec = host_feed_rehearsal(torch.device("cuda", local_rank), 768, args.workload, args.dataset_clips, args.clips_per_step, rank, world,
                                  args.stub_compute_ms, prefetch=args.prefetch, workers=args.loader_workers, n_resident=args.resident_clips or 4)
        if rank == 0:
            print(json.dumps(rec))
        if world > 1:
            rdist.barrier()
            dist.destroy_process_group()
        return
    eng = RelaxEngine(local_rank)
    rn_sd = synth.resnet50_state_dict()
    vit_sd = synth.vit_state_dict("vit_base")
    eng.load_resnet50(rn_sd)
    eng.load_vit(vit_sd, "vit_base")       # every workload but config 2 needs it (other_workloads included)
    B = args.clips_per_step
    eng.reserve(2 * T * B)
    eng.set_precision(args.precision)
    eng.set_option("gemm_split_k", args.gemm_split_k)
    precision = eng.precision()          # what the ENGINE computes in (read back from the library, not the flag)
    assert precision == args.precision, (precision, args.precision)
    x3, h2 = precision == "bf16x3", precision == "f16x2"
    x6 = precision in ("bf16x6", "f16x2")      # the split-operand kernels (under f16x2 only the launches without an f16x2 kernel stay bf16x6)
    if h2:
        assert eng.get_option("h2_form") == 1  # (three products for K >= 256: every f16x2 launch of these workloads; H2_EXECUTED below)
        assert eng.get_option("rn_h2") == 1 and eng.get_option("rn_h2_early") == 1 and eng.get_option("att_h2") == 1

    def barrier():
        if world > 1:
            rdist.barrier()
        torch.cuda.synchronize()

    if args.dataset_clips:
        dataset_mode(args, eng, rank, world, barrier, precision)
        return
    if args.traffic_child:      # the PMC target of measure_hbm_traffic: the same step function, one warm-up + one step, no JSON
        n_res = args.resident_clips or 2
        res_ = [torch.from_numpy(synth.synthetic_clip(T, H, W, clip_id=i, distinct=4)).cuda() for i in range(n_res)]
        for i in range(2):
            batch = [res_[(i * B + j) % n_res] for j in range(B)]
            if args.workload.startswith("full"):
                eng.full_clip_vectors(batch, flow=True)
            else:
                eng.clip_vectors(batch, resnet=True, vit=use_vit)
        torch.cuda.synchronize()
        return

    def make_step(workload, clips_per_step, n_resident=2, seed_base=0, distinct=4):
        """-> (step(i), feature dim).  Two distinct resident clips per rank, alternated (inputs are in HBM before timing starts)."""
        h_, w_, t_, vit_ = WORKLOADS[workload]
        resident = [torch.from_numpy(synth.synthetic_clip(t_, h_, w_, clip_id=seed_base + rank * n_resident + i, distinct=distinct)).cuda()
                    for i in range(n_resident)]
        is_full = workload.startswith("full")

        def step(i):
            batch = [resident[(i * clips_per_step + j) % n_resident] for j in range(clips_per_step)]
            if is_full:
                vecs = eng.full_clip_vectors(batch, flow=True)
            else:
                vecs = eng.clip_vectors(batch, resnet=True, vit=vit_)          # [B, feat_dim]
            if world > 1:
                return rdist.gather_clip_vectors(vecs, world * clips_per_step, rank, world)
            return vecs
        return step, (35203 if is_full else 15171 + (4608 if vit_ else 0)), resident

    def timed(step, steps, warmup):
        """warmup untimed steps, then exactly `steps` steps between two barrier + synchronize brackets; the contraction / fragment /
        flow kernels of the timed steps are also timed one by one with HIP events on the launch stream."""
        for i in range(warmup):
            step(i)
        barrier()
        eng.profile_enable(True)
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            out = step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        prof = {"gemm": eng.profile_read(3 if x6 else 0), "frag": eng.profile_read(1), "gemm_bytes": eng.profile_read(4 if x6 else 2),
                "flow": eng.profile_read(5), "flow_stage": eng.profile_read(6), "other": eng.profile_read(0 if x6 else 3),
                "h2": eng.profile_read(7), "h2_bytes": eng.profile_read(8),
                "dom": eng.profile_read(9), "dom_bytes": eng.profile_read(10)}
        eng.profile_enable(False)
        return elapsed, out, prof

    n_resident = args.resident_clips or 2
    step, feat_dim, clips = make_step(args.workload, B, n_resident)
    full = args.workload.startswith("full")
    elapsed, out, prof = timed(step, args.steps, args.warmup)
    x6_ms, x6_flops, x6_launches = prof["gemm"]                 # bf16x6 launches (fp32 / bf16x3: the launches of that kernel)
    h2_ms, h2_flops, h2_launches = prof["h2"]                   # f16x2 launches (none under the other precisions)
    gemm_ms, gemm_flops, gemm_launches = x6_ms + h2_ms, x6_flops + h2_flops, x6_launches + h2_launches
    frag_ms, frag_bytes, frag_launches = prof["frag"]
    gemm_alg_bytes = prof["gemm_bytes"][1] + prof["h2_bytes"][1]
    flow_ms, flow_bytes, flow_launches = prof["flow"]
    other_ms, other_flops, other_launches = prof["other"]   # contraction launches on the other kernel family
    assert eng.precision() == precision
    assert out.shape == (world * B, feat_dim) and bool(torch.isfinite(out).all())

    # metric (ii) of SURVEY §8(d): the same steps with every clip copied from pinned host memory on a side stream
    h2d = None
    if not args.no_h2d and not full:      # (at every N: each rank feeds its own GPU from its own pinned clips, the all-gather stays in the step)
        from relax_vqa_amd.feeder import PinnedClipFeeder
        host_clips = [c.cpu().pin_memory() for c in clips]
        feeder = PinnedClipFeeder([host_clips[j % n_resident] for j in range(B)], B, eng.device)

        def run(batch):
            vecs = eng.clip_vectors(batch, resnet=True, vit=use_vit)
            return rdist.gather_clip_vectors(vecs, world * B, rank, world) if world > 1 else vecs

        feeder.run(2, run)
        barrier()
        t2 = time.perf_counter()
        feeder.run(args.steps, run)     # `steps` copies and `steps` compute passes inside the clock: the first copy has nothing to hide under
        barrier()
        e2 = time.perf_counter() - t2
        if world > 1:
            e2 = rdist.all_reduce_max(e2, "cuda")
        h2d = {"value": args.steps * world * B / e2, "unit": "clips/s", "ms_per_step": e2 / args.steps * 1e3,
               "note": "every clip copied pinned host -> device on a side stream, double-buffered under the compute; all `steps` copies are "
                       "inside the timed region (the first one is exposed, the others hide under the previous step)"}
        del feeder, host_clips

    # the other fp32-grade arithmetic, measured beside the headline on the same workload and step function
    fast = None
    other = "bf16x6" if h2 else ("fp32" if x6 else "bf16x6")
    if world == 1 and not args.no_fast_mode and not x3:
        eng.set_precision(other)
        for i in range(2):
            step(i)
        barrier()
        eng.profile_enable(True)
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        e1 = time.perf_counter() - t1
        f_ms, f_flops, f_n = eng.profile_read(0 if other == "fp32" else 3)
        eng.profile_enable(False)
        eng.set_precision(precision)
        fast = {"precision": "exact fp32 products on v_mfma_f32_32x32x2_f32 (an fp32 FMA chain)" if other == "fp32" else
                             "bf16x6: three bf16 planes per fp32 operand, six partial products on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, two products per instruction; 32x32x16 on the narrow tiles), fp32 accumulate",
                "value": args.steps * B / e1, "unit": "clips/s", "ms_per_step": e1 / args.steps * 1e3,
                "contraction_algorithmic_tflops": f_flops / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0,
                "frac_of_fp32_mfma_peak": (f_flops / (f_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS) if (f_ms > 0 and other == "fp32") else None}

    # the other BASELINE configurations on this GPU, a few steps each (the headline stays config 3)
    others = None
    if world == 1 and not args.no_other_workloads and args.workload == "config3" and x6:   # (bf16x6 or f16x2)
        del clips, step
        others = {}
        for name, b_o in (("config2", 64), ("config4", 64), ("full2160p", 8)):
            torch.cuda.empty_cache()
            step_o, dim_o, clips_o = make_step(name, b_o, 2, seed_base=50, distinct=2 if name == "full2160p" else 4)
            e_o, out_o, prof_o = timed(step_o, 3, 1)
            assert out_o.shape == (b_o, dim_o) and bool(torch.isfinite(out_o).all())
            g_ms, g_flops, g_n = prof_o["gemm"]
            q_ms, q_flops, q_n = prof_o["h2"]
            t_ms = g_ms + q_ms
            rec = {"value": 3 * b_o / e_o, "unit": "clips/s", "ms_per_step": e_o / 3 * 1e3, "clips_per_step": b_o, "steps": 3,
                   "feature_dim": dim_o,
                   "roofline": contraction_roofline(g_ms, g_flops, q_ms, q_flops, 6.0, e_o, prof_o.get("dom"))}
            fs = flow_stage_record(prof_o, e_o)
            if fs is not None:
                rec["roofline_flow_stage"] = fs
            others[name] = rec
            del step_o, clips_o, out_o
        torch.cuda.empty_cache()
        others["config4_dataset"] = dataset_pass(eng, "config4", 512, 64, 0, 1, args.gemm_split_k, host_clips=True, prefetch=args.prefetch,
                                                 workers=args.loader_workers, n_resident=4, warmup=1)

    if world > 1:
        elapsed = rdist.all_reduce_max(elapsed, "cuda")

    traffic, traffic_note = hbm_traffic_per_launch(args.workload, B, precision), (
        "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes, FETCH doubled per the gfx950 guide), from the "
        "committed profile of this workload, batch and precision: profiles/r05_hbm_traffic.json (null when none matches the run)")
    flow_traffic = None
    if world == 1 and not args.no_measure_traffic and x6:
        torch.cuda.synchronize()
        live, note, flow_traffic = measure_hbm_traffic(args.workload, B, precision, args.gemm_split_k)
        if live is not None:
            traffic, traffic_note = live, note
        else:
            traffic_note += f"; the live measurement was not possible ({note})"

    if rank == 0:
        clips_total = args.steps * world * B
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        mult = {"fp32": 1.0, "bf16x3": 3.0, "bf16x6": 6.0, "f16x2": 6.0}[precision]
        executed = (mult * x6_flops + H2_EXECUTED * h2_flops) / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0   # matrix-pipe FLOPs per second
        result = {
            "metric": "clips/sec (32 sampled frames, 1080p) feature extraction" if args.workload == "config3"
                      else f"clips/sec feature extraction ({args.workload})",
            "value": clips_total / elapsed, "unit": "clips/s", "n_gpus": world, "ranks": world,
            "backend": dist.get_backend() if world > 1 else None,
            "rccl_ranks": _rccl_ranks(world, dist.get_backend() if world > 1 else None), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[precision],
            "data": "synthetic",
            "config": {"workload": workload_text(args.workload), "clips_per_step_per_gpu": B, "pairs_per_clip": T,
                       "feature_dim": feat_dim, "parallelism": f"clip-sharded dp{world}, RCCL all-gather of per-clip vectors"},
            "roofline": headline_roofline(precision, x6, x6_ms, x6_flops, x6_launches, h2_ms, h2_flops, h2_launches, mult, gemm_alg_bytes, elapsed,
                                          prof["dom"], prof["dom_bytes"], traffic, traffic_note, other_ms, other_launches, args.steps),
            "roofline_fragment_stage": {
                "bound": "hbm", "kernel": "patch_score_aligned<pair> (fused absdiff + 16x16 patch sums)",
                "achieved": frag_bytes / (frag_ms * 1e-3) / 1e9 if frag_ms > 0 else 0.0, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": (frag_bytes / (frag_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if frag_ms > 0 else 0.0,
                "traffic": None, "launches": frag_launches,
            },
        }
        if full and flow_launches:
            fs = flow_stage_record(prof, elapsed, flow_traffic)
            if fs is not None:
                result["roofline_flow_stage"] = fs
        if fast is not None:
            result["exact_fp32_mode" if other == "fp32" else "bf16x6_mode"] = fast
        if h2d is not None:
            result["with_pinned_host_to_device_copy"] = h2d
        if others is not None:
            result["other_workloads"] = others
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(H, W, T, use_vit, rn_sd, vit_sd if use_vit else None, args.cpu_sample_pairs)
            # (two significant digits: the CPU figure moves 0.06 - 0.10 clips/s by box, and a large GPU / CPU ratio says nothing about
            # kernel quality - the roofline fraction does)
            ratio = result["value"] / result["cpu_baseline"]["value"]
            result["speedup_vs_cpu_faithful"] = float(f"{ratio:.2g}")
        print(json.dumps(result))
    if world > 1:
        rdist.barrier()
        dist.destroy_process_group()


KERNEL_TEXT = {
    "fp32": "conv_gemm_f32 (fp32 implicit-GEMM conv / GEMM, v_mfma_f32_32x32x2_f32)",
    "bf16x3": "conv_gemm_f32<..., bf16x3> (3 x v_mfma_f32_32x32x16_bf16 per fp32 product)",
    "bf16x6": "gemm_x6 + conv1_x6 (six bf16 partial products per fp32 product on v_mfma_f32_16x16x32_bf16 / 32x32x16)",
    "f16x2": "every contraction launch: gemm_h3 (f16x2: three fp16 partial products per fp32 product on v_mfma_f32_16x16x32_f16 - the ViT GEMMs, ResNet-50's "
             "layer3 / layer4, layer2[0]'s downsample), the f16x2 kernels of ResNet-50's stem, layer1 and layer2 (v_mfma_f32_32x32x16_f16: 3x3 + conv3 back to back, "
             "layer2's conv1) and what is left on gemm_x6 (bf16x6, six products: layer1's three conv1 launches)"}
FRAC_NOTE = ("frac = achieved / peak with achieved = ALGORITHMIC fp32 FLOPs (2 x MAC of the contraction) / launch time (SURVEY 8(d)); pipe_busy_frac = the 16-bit "
             "matrix-pipe FLOPs actually executed (3 per fp32 product under f16x2, 6 under bf16x6) / the same peak: how busy the pipe is, not what the kernel delivers")


def contraction_roofline(x6_ms, x6_flops, h2_ms, h2_flops, x6_mult, elapsed_s, dom=None, peak=None):
    """{bound, achieved, peak, frac, ...} of all contraction launches of a timed region: achieved = algorithmic TFLOP/s over their summed
    launch time; the executed (pipe) rate beside it."""
    peak = peak or BF16_MATRIX_PEAK_TFLOPS
    t_ms = x6_ms + h2_ms
    alg = (x6_flops + h2_flops) / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
    exe = (x6_mult * x6_flops + H2_EXECUTED * h2_flops) / (t_ms * 1e-3) / 1e12 if t_ms > 0 else 0.0
    rec = {"bound": "mfma", "kernel": "all contraction launches (f16x2 + bf16x6 families)" if h2_ms > 0 else "all contraction launches", "unit": "TFLOP/s", "peak": peak,
           "achieved": alg, "frac": alg / peak, "executed_tflops": exe, "pipe_busy_frac": exe / peak,
           "kernel_time_share_of_step": t_ms * 1e-3 / elapsed_s}
    if dom is not None and dom[2]:
        rec["dominant_kernel"] = dominant_record(dom, None, elapsed_s, None)
    return rec


def dominant_record(dom, dom_bytes, elapsed_s, traffic):
    """gemm_h3<false,false,false> alone (span kind 6: the plain f16x2 GEMMs = the ViT's): recomputable from profiles/*_kernel_stats.csv."""
    d_ms, d_flops, d_n = dom
    alg = d_flops / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
    rec = {"kernel": "gemm_h3<false,false,false> (the plain f16x2 GEMM: the ViT's 49 GEMMs per pass, three fp16 products per fp32 product)",
           "launches": d_n, "avg_launch_us": d_ms * 1e3 / max(d_n, 1), "algorithmic_gflop_per_launch": d_flops / max(d_n, 1) / 1e9,
           "achieved": alg, "unit": "TFLOP/s", "peak": BF16_MATRIX_PEAK_TFLOPS, "frac": alg / BF16_MATRIX_PEAK_TFLOPS,
           "executed_tflops": H2_EXECUTED * alg, "pipe_busy_frac": H2_EXECUTED * alg / BF16_MATRIX_PEAK_TFLOPS,
           "time_share_of_step": d_ms * 1e-3 / elapsed_s}
    if dom_bytes is not None:
        rec["algorithmic_bytes_per_launch"] = dom_bytes[1] / max(d_n, 1)
    if traffic is not None:
        rec["traffic"] = traffic[0]
        rec["traffic_dispatches_in_pmc_pass"] = traffic[1]
        if dom_bytes is not None and dom_bytes[1] > 0:
            rec["traffic_over_algorithmic"] = traffic[0] / (dom_bytes[1] / max(d_n, 1))
    return rec


def headline_roofline(precision, x6, x6_ms, x6_flops, x6_launches, h2_ms, h2_flops, h2_launches, mult, alg_bytes, elapsed_s, dom, dom_bytes,
                      traffic, traffic_note, other_ms, other_launches, steps):
    """The bench line's `roofline`.  Under f16x2 with a ViT in the workload the dominant kernel is gemm_h3<false,false,false> and the
    top-level figures are THAT kernel's (algorithmic FLOPs per launch / its average launch time, measured live with HIP events on the
    launch stream); `all_contractions` holds the aggregate over every contraction launch of both families."""
    peak = FP32_MATRIX_PEAK_TFLOPS if precision == "fp32" else BF16_MATRIX_PEAK_TFLOPS
    allc = contraction_roofline(x6_ms, x6_flops, h2_ms, h2_flops, mult, elapsed_s, None, peak)
    n_all = x6_launches + h2_launches
    allc.update({"kernel": KERNEL_TEXT[precision], "launches": n_all, "avg_launch_us": (x6_ms + h2_ms) * 1e3 / max(n_all, 1),
                 "algorithmic_gflop_per_launch": (x6_flops + h2_flops) / max(n_all, 1) / 1e9,
                 "algorithmic_bytes_per_launch": alg_bytes / max(n_all, 1), "traffic": traffic, "traffic_note": traffic_note,
                 "families": {"f16x2": {"launches": h2_launches, "ms_per_step": h2_ms / steps,
                                        "algorithmic_tflops": h2_flops / (h2_ms * 1e-3) / 1e12 if h2_ms > 0 else None, "executed_per_algorithmic": H2_EXECUTED},
                              "bf16x6" if x6 else precision: {"launches": x6_launches, "ms_per_step": x6_ms / steps,
                                                              "algorithmic_tflops": x6_flops / (x6_ms * 1e-3) / 1e12 if x6_ms > 0 else None,
                                                              "executed_per_algorithmic": mult}},
                 "other_contraction_kernel": {"ms_per_step": other_ms / steps, "launches": other_launches}})
    if dom[2] and dom[0] > 0.5 * (x6_ms + h2_ms):       # one kernel holds most of the contraction time: the line is about it
        rec = dominant_record(dom, dom_bytes, elapsed_s, DOMINANT_TRAFFIC)
        rec["bound"] = "mfma"
        rec.setdefault("traffic", None)
        rec["frac_note"] = FRAC_NOTE
        rec["kernel_time_share_of_step"] = rec["time_share_of_step"]
        rec["all_contractions"] = allc
        return rec
    allc["frac_note"] = FRAC_NOTE
    return allc


FLOW_KERNEL = ("flow_iteration (one Farneback iteration per launch: matrix entries from the two polynomial expansions and the flow, 15x15 box "
               "filter, 2x2 solve; 56 algorithmic bytes per pixel)")


def flow_stage_record(prof, elapsed_s, traffic_per_clip=None):
    """roofline_flow_stage: the WHOLE Farneback stage (pyramid, polynomial expansion, 4 levels x 3 iterations, visualisation) against
    HBM: achieved = the algorithmic bytes of all its kernels (inputs read once, outputs written once per kernel) / the stage's time,
    first launch to last (HIP events on the launch stream); `dominant_kernel` = the iteration kernel alone."""
    st_ms, st_bytes, st_n = prof["flow_stage"]
    it_ms, it_bytes, it_n = prof["flow"]
    if not st_n or st_ms <= 0:
        return None
    rec = {"bound": "hbm", "kernel": "the whole Farneback stage: pyramid_fused, poly_expansion, flow_iteration x 12, flow_visualise",
           "achieved": st_bytes / (st_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": st_bytes / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "algorithmic_bytes_per_stage": st_bytes / st_n, "stages": st_n, "avg_stage_ms": st_ms / st_n,
           "traffic": traffic_per_clip,
           "traffic_over_algorithmic": (traffic_per_clip / (st_bytes / st_n)) if traffic_per_clip else None,
           "time_share_of_step": st_ms * 1e-3 / elapsed_s,
           "dominant_kernel": {"kernel": FLOW_KERNEL, "achieved": it_bytes / (it_ms * 1e-3) / 1e9 if it_ms > 0 else 0.0, "unit": "GB/s",
                               "frac": (it_bytes / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if it_ms > 0 else 0.0,
                               "algorithmic_bytes_per_pixel_level_iteration": 56, "launches": it_n,
                               "avg_launch_us": it_ms * 1e3 / max(it_n, 1), "time_share_of_step": it_ms * 1e-3 / elapsed_s}}
    return rec
H2_EXECUTED = 3.0   # fp16 MFMA products per fp32 product in gemm_h3 at K >= 256 ("h2_form" 1): A[lo] B[hi], A[hi] B[lo], A[hi] B[hi]
DTYPE_TEXT = {"f16x2": "f32 (fp32-grade split-operand arithmetic, fp32 accumulate: the whole ViT (GEMMs and attention), ResNet-50's stem, layer3 / layer4, "
                       "and in layer1 / layer2 the 3x3 convolutions, every conv3 (back to back with its 3x3), the downsample convolutions and layer2's conv1 on fp32 operands "
                       "as 2 fp16 planes x a power-of-two scale, 3 partial products (4 below K = 256) on the fp16 MFMA [f16x2]; the three conv1 (1x1) launches of "
                       "layer1 (HBM-bound) on 3 bf16 planes, 6 partial products on the bf16 MFMA [bf16x6])",
              "fp32": "f32", "bf16x3": "bf16x3 (fp32 operands split into two bf16 terms, fp32 accumulate; reduced precision)",
              "bf16x6": "f32 (fp32 operands as 3 bf16 planes, 6 partial products on the bf16 MFMA, fp32 accumulate: fp32-grade)"}


def workload_text(name):
    H, W, T, use_vit = WORKLOADS[name]
    return (f"{name}: synthetic {W}x{H} clips, {T} (frame,next) pairs, residual fragments + ResNet-50 layer-stack/pool"
            + (" + ViT-B/16 pool" if use_vit else "")
            + (" + whole-frame features + Farneback flow fragments (35203-d)" if name.startswith("full") else "") + ", random-init weights")


class StubEngine:
    """The engine of the host-feed rehearsal: clip_vectors reads one line of every clip on the device (so the clip's copy must have
    landed) and then holds the stream for `ms` milliseconds; returns zeros.  Same call signature and device as RelaxEngine."""

    def __init__(self, device, vit_dim, ms):
        self.device, self.vit_dim, self.ms = device, vit_dim, ms
        torch.cuda._sleep(1000)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        torch.cuda._sleep(20_000_000)
        e.record()
        torch.cuda.synchronize()
        self.cycles_per_ms = 20_000_000 / s.elapsed_time(e)

    def clip_vectors(self, clips, resnet=True, vit=True, per_frame=False):
        acc = torch.zeros((), dtype=torch.int64, device=self.device)
        for c in clips:                                                        # every clip is read on the device, in stream order
            acc += c.reshape(-1)[:: max(c.numel() // 4096, 1)].sum()
        torch.cuda._sleep(int(self.ms * self.cycles_per_ms))
        F = (13120 + 2051 if resnet else 0) + (6 * self.vit_dim if vit else 0)
        return torch.zeros((len(clips), F), dtype=torch.float32, device=self.device) + (acc * 0).float()


def host_feed_rehearsal(device, vit_dim, workload, n, B, rank, world, stub_ms, prefetch=2, workers=8, n_resident=4):
    """Eight (or `world`) ranks' HOST side at once: every rank runs the sharded dataset pass over clips that live in pageable host
    memory - loader threads, pinned pool, copy stream, as in production - against a stand-in engine that waits `stub_ms` per batch.
    On a one-GPU box (RELAX_DIST_BACKEND=gloo) all ranks share the device and its ONE PCIe link, so the H2D figure is the rate of
    that link shared by the ranks; the staging figure (pageable -> pinned, on the host's cores and memory) and the pinned totals are
    what an 8-GPU node's host would see.  -> the record (rank 0 prints it)."""
    from relax_vqa_amd import dataset
    H, W, T, use_vit = WORKLOADS[workload]
    host = [synth.synthetic_clip(T, H, W, clip_id=700 + i, distinct=2) for i in range(n_resident)]
    stub = StubEngine(device, vit_dim, stub_ms)
    kw = dict(clips_per_step=B, resnet=True, vit=use_vit, rank=rank, world=world, prefetch=prefetch, workers=workers, batch_invariant=False)

    def sync():
        if world > 1:
            rdist.barrier()
        torch.cuda.synchronize()

    def timed_pass(source):
        dataset.extract_dataset_clips(source, min(n, (prefetch + 2) * B * world), stub, **kw)     # warm-up: pins the pool
        sync()
        t = {}
        t0 = time.perf_counter()
        _, errors = dataset.extract_dataset_clips(source, n, stub, timings=t, **kw)
        sync()
        t["wall_s"] = time.perf_counter() - t0
        assert not errors, errors[:3]
        return t

    clip_bytes = host[0].nbytes
    copy = timed_pass(lambda i: host[i % n_resident])                    # pageable source: staged by a loader-thread copy
    def decode_into(i, alloc):                                           # `alloc` protocol: "decoded" straight into pinned memory
        out = alloc(host[i % n_resident].shape)
        np.copyto(out, host[i % n_resident])
        return out
    direct = timed_pass(decode_into)
    per_rank = {"rank": rank, "clips": len(rdist.shard_clips(n, rank, world)),
                "copy": {k: copy[k] for k in ("wall_s", "loader_wait_s", "h2d_bytes", "staged_bytes", "pinned_peak_bytes", "pinned_live_bytes",
                                             "pinned_pool_limit_bytes", "loader_cpus")},
                "direct": {k: direct[k] for k in ("wall_s", "loader_wait_s", "h2d_bytes", "staged_bytes", "pinned_peak_bytes", "pinned_live_bytes")}}
    allr = rdist.gather_objects(per_rank, world)
    wall_c = max(r["copy"]["wall_s"] for r in allr)
    wall_d = max(r["direct"]["wall_s"] for r in allr)
    need = {"config3": 101.7, "config4": 200.0, "config2": 600.0}.get(workload, 0.0) * clip_bytes / 1e9     # GB/s per GPU at this round's rates
    return {
        "what": f"host-feed REHEARSAL, not a feature-extraction rate: {world} ranks ({dist.get_backend() if world > 1 else 'single process'}) sharing "
                f"{torch.cuda.device_count()} GPU(s); backbones replaced by a {stub_ms} ms device-side wait per batch of {B} clips",
        "workload": workload_text(workload), "ranks": world, "gpus_on_box": torch.cuda.device_count(), "clips": n, "clip_MB": clip_bytes / 1e6,
        "clips_per_step_per_rank": B, "prefetch_batches": prefetch, "loader_workers_per_rank": workers,
        "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
        "staged_through_a_copy": {"aggregate_clips_per_s": n / wall_c, "aggregate_staging_GBps": sum(r["copy"]["staged_bytes"] for r in allr) / wall_c / 1e9,
                                  "aggregate_h2d_GBps": sum(r["copy"]["h2d_bytes"] for r in allr) / wall_c / 1e9,
                                  "loader_wait_s_max": max(r["copy"]["loader_wait_s"] for r in allr), "wall_s": wall_c},
        "decoded_into_pinned": {"aggregate_clips_per_s": n / wall_d, "aggregate_staging_GBps": sum(r["direct"]["staged_bytes"] for r in allr) / wall_d / 1e9,
                                "aggregate_h2d_GBps": sum(r["direct"]["h2d_bytes"] for r in allr) / wall_d / 1e9,
                                "loader_wait_s_max": max(r["direct"]["loader_wait_s"] for r in allr), "wall_s": wall_d},
        "stub_ceiling_clips_per_s": world * B / (stub_ms * 1e-3),
        "need_per_gpu_GBps": need, "need_node_GBps": need * world,
        "pinned_GB_total_peak": sum(max(r["copy"]["pinned_peak_bytes"], r["direct"]["pinned_peak_bytes"]) for r in allr) / 2 ** 30,
        "pinned_GB_total_kept": sum(r["direct"]["pinned_live_bytes"] for r in allr) / 2 ** 30,
        "pinned_pool_limit_GB_per_rank": allr[0]["copy"]["pinned_pool_limit_bytes"] / 2 ** 30,
        "loader_cpus_bound_per_rank": [r["copy"]["loader_cpus"] for r in allr],
        "note": "one GPU on this box: the H2D rate is ONE PCIe link shared by all ranks (an 8-GPU node has eight); the staging rate and the "
                "pinned totals are host-side and carry over"}


def write_frame_files(directory, n_videos, T, H, W):
    """Sampled-frame PNGs of n_videos synthetic videos under `directory`, as the reference's ffmpeg step leaves them
    (src/video_frames_extract.py:51-69: {video}_{n}.png, {video}_{n}_next.png).  Content: low-pass noise + fine noise (compresses about
    like camera footage; pure noise would be the PNG decoder's easiest case).  Skipped if the files are there."""
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(11)
    for v in range(n_videos):
        name = f"video{v}"
        if os.path.exists(os.path.join(directory, f"{name}_{T - 1}_next.png")):
            continue
        for t in range(T):
            coarse = rng.integers(0, 256, (H // 16 + 2, W // 16 + 2, 3), dtype=np.uint8)
            base = np.asarray(Image.fromarray(coarse).resize((W, H), Image.BICUBIC)).astype(np.int16)
            for suffix in ("", "_next"):
                frame = np.clip(base + rng.integers(-6, 7, base.shape, dtype=np.int16), 0, 255).astype(np.uint8)
                Image.fromarray(frame).save(os.path.join(directory, f"{name}_{t}{suffix}.png"), compress_level=3)
    return [f"video{v}" for v in range(n_videos)]


def dataset_pass(eng, workload, n, B, rank, world, split_k, host_clips=False, prefetch=2, workers=8, n_resident=4, warmup=1, dump=None,
                 frame_files=None, workers_sweep=None, loader_pools=()):
    """BASELINE config 4 as written: n clips sharded over the ranks (relax-vqa_amd/dataset.py), ONE all-gather of the [n, F] matrix;
    strong scaling: value = n clips / the time of the whole pass (max over ranks), warm-up batches untimed.  `value` is the pass over
    device-resident clips (the metric's definition); host_clips adds the same pass fed from pageable host memory.
    -> the record (a dict; complete on every rank, printed by rank 0)."""
    from relax_vqa_amd import dataset
    H, W, T, use_vit = WORKLOADS[workload]
    full = workload.startswith("full")
    # every rank holds the same few distinct clips; clip i of the list is resident[i % n_resident] (the list is synthetic: what is
    # measured is the pass over n clips, and a rank count must not change which pixels clip i has)
    host = [synth.synthetic_clip(T, H, W, clip_id=700 + i, distinct=4) for i in range(n_resident)]
    resident = [torch.from_numpy(c).cuda() for c in host]
    F = dataset.feature_dim(eng, True, use_vit, full)
    # batch_invariant: the pass itself switches the tail split-K off unless the command line asks for the split (rank-count-invariant bits)
    kw = dict(clips_per_step=B, resnet=True, vit=use_vit, full=full, rank=rank, world=world, prefetch=prefetch, workers=workers,
              batch_invariant=(split_k == 0))

    def barrier():
        if world > 1:
            rdist.barrier()
        torch.cuda.synchronize()

    def one_pass(source, warm_clips, ramp):
        for _ in range(warmup):
            dataset.extract_dataset_clips(source, min(warm_clips, n), eng, ramp=False, **kw)
        barrier()
        timings = {}
        t0 = time.perf_counter()
        matrix, errors = dataset.extract_dataset_clips(source, n, eng, timings=timings, ramp=ramp, **kw)
        barrier()
        elapsed = time.perf_counter() - t0
        gather_s = timings["all_gather_s"]
        if world > 1:
            elapsed = rdist.all_reduce_max(elapsed, "cuda")
            gather_s = rdist.all_reduce_max(gather_s, "cuda")
        assert matrix.shape == (n, F) and not errors and bool(torch.isfinite(matrix).all()), (matrix.shape, errors[:3])
        return matrix, elapsed, gather_s, timings

    matrix, elapsed, gather_s, timings = one_pass(lambda i: resident[i % n_resident], B * world, False)
    if dump and rank == 0:
        np.save(dump, matrix.cpu().numpy())
    per_rank = -(-n // world)
    backend = dist.get_backend() if world > 1 else None
    rec = {
        "metric": f"clips/sec feature extraction, dataset pass ({workload}, {n} clips sharded over the ranks)",
        "value": n / elapsed, "unit": "clips/s", "n_gpus": world, "ranks": world, "backend": backend,
        "rccl_ranks": _rccl_ranks(world, backend), "steps": -(-per_rank // B),
        "warmup": warmup, "ms_per_step": elapsed / (-(-per_rank // B)) * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": DTYPE_TEXT[eng.precision()], "data": "synthetic",
        "config": {"workload": workload_text(workload) + f"; {n} clips, contiguous shards of <= {per_rank}",
                   "dataset_clips": n, "clips_per_step_per_gpu": B, "pairs_per_clip": T, "feature_dim": F, "distinct_clips": n_resident,
                   "parallelism": f"clip-sharded dp{world}, one RCCL all-gather of the [{n}, {F}] matrix"},
        "all_gather_ms": gather_s * 1e3, "extract_s": timings["extract_s"], "errors": 0,
        "gemm_split_k": split_k, "prefetch_batches": prefetch, "loader_workers": workers}
    if host_clips:
        # the same pass with every clip in pageable host memory (numpy arrays, as a decoder would hand them over)
        m_h, e_h, g_h, t_h = one_pass(lambda i: host[i % n_resident], (prefetch + 2) * B * world, True)
        # (the host-fed pass opens with three short batches: with the tail split-K on, rows of those clips differ in the last bits)
        if split_k == 0:
            assert torch.equal(m_h, matrix), "the host-fed pass changed the matrix"
        else:
            assert torch.allclose(m_h, matrix, rtol=1e-4, atol=1e-6), "the host-fed pass changed the matrix"
        # what the copies would cost alone: one batch of this rank's clips, pinned -> device, timed on an idle GPU
        pin = [torch.from_numpy(host[j % n_resident]).pin_memory() for j in range(min(B, 8))]
        dst = [torch.empty_like(p, device="cuda") for p in pin]
        dst[0].copy_(pin[0], non_blocking=True)          # untimed: the first copy from freshly pinned pages was seen 10 x slower on one box
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for p, d in zip(pin, dst):
            d.copy_(p, non_blocking=True)
        torch.cuda.synchronize()
        bw = sum(p.numel() for p in pin) / (time.perf_counter() - tc)
        t_copy = t_h["h2d_bytes"] / bw
        rec["host_fed"] = {"value": n / e_h, "unit": "clips/s", "frac_of_device_resident": elapsed / e_h,
                           "h2d_hidden_frac": max(0.0, min(1.0, 1.0 - max(0.0, e_h - elapsed) / t_copy)) if t_copy > 0 else None,
                           "h2d_GB": t_h["h2d_bytes"] / 1e9, "h2d_alone_s": t_copy, "pinned_h2d_GBps": bw / 1e9,
                           "loader_wait_s": t_h["loader_wait_s"], "extract_s": t_h["extract_s"],
                           "matrix_equal_to_device_resident": "bit for bit" if split_k == 0 else "to 1e-4 (tail split-K on: bits follow the batch composition)",
                           "note": "clips in pageable host memory -> loader threads copy them into pinned staging -> side-stream H2D of batch "
                                   "k+1 under the compute of batch k (relax-vqa_amd/dataset.py::ClipStager); never `value`"}
        del pin, dst
    if frame_files:
        # the same pass with every clip read from sampled-frame PNG files by the loader threads (sampling.load_clip_from_frames, decoded
        # straight into pinned staging memory): where a from-files run saturates, by loader-thread count
        from relax_vqa_amd import sampling
        names = write_frame_files(frame_files, n_resident, T, H, W) if rank == 0 else None
        barrier()
        names = [f"video{v}" for v in range(n_resident)]
        size = sum(os.path.getsize(os.path.join(frame_files, f)) for f in os.listdir(frame_files) if f.startswith("video0_")) / (2 * T)
        sweep = []
        for w in (workers_sweep or [workers]):
            kw_w = dict(kw, workers=w)
            src = lambda i, alloc=None: sampling.load_clip_from_frames(frame_files, names[i % n_resident], alloc=alloc)   # noqa: E731
            dataset.extract_dataset_clips(src, min(n, 2 * B * world), eng, ramp=False, **kw_w)        # warm-up (page cache, pinned pool)
            barrier()
            t_f = {}
            t0 = time.perf_counter()
            m_f, err_f = dataset.extract_dataset_clips(src, n, eng, timings=t_f, ramp=True, **kw_w)
            barrier()
            e_f = time.perf_counter() - t0
            if world > 1:
                e_f = rdist.all_reduce_max(e_f, "cuda")
            assert not err_f and bool(torch.isfinite(m_f).all()), err_f[:3]
            sweep.append({"loader_workers_per_rank": w, "value": n / e_f, "unit": "clips/s", "frac_of_device_resident": elapsed / e_f,
                          "png_decodes_per_s": n * 2 * T / e_f, "png_decodes_per_s_per_loader_thread": n * 2 * T / e_f / (w * world),
                          "loader_wait_s": t_f["loader_wait_s"], "staged_bytes": t_f["staged_bytes"], "loader_cpus_bound": t_f["loader_cpus"]})
        # ... and by loader PROCESSES (relax-vqa_amd/loaderpool.py): the decode leaves this process (and its GIL); P workers write the clips
        # into shared memory this rank has page-locked, the driver's loader threads only wait for them
        proc_sweep = []
        for p_, pool in loader_pools:
            kw_p = dict(kw, workers=max(2 * p_, 8))
            dataset.extract_dataset_clips(pool, min(n, 2 * B * world), eng, ramp=False, **kw_p)       # warm-up (page cache, segments, registration)
            barrier()
            t_f = {}
            t0 = time.perf_counter()
            m_p, err_p = dataset.extract_dataset_clips(pool, n, eng, timings=t_f, ramp=True, **kw_p)
            barrier()
            e_f = time.perf_counter() - t0
            if world > 1:
                e_f = rdist.all_reduce_max(e_f, "cuda")
            assert not err_p and bool(torch.isfinite(m_p).all()), err_p[:3]
            same = bool(torch.equal(m_p, m_f)) if sweep else None      # against the last thread-path matrix (same ramp, same batches)
            proc_sweep.append({"loader_processes_per_rank": p_, "value": n / e_f, "unit": "clips/s", "frac_of_device_resident": elapsed / e_f,
                               "png_decodes_per_s": n * 2 * T / e_f, "png_decodes_per_s_per_process": n * 2 * T / e_f / (p_ * world),
                               "loader_wait_s": t_f["loader_wait_s"], "staged_bytes": t_f["staged_bytes"],
                               "matrix_equal_to_thread_path": same})
            pool.close()
        rec["from_frame_files"] = {"sweep": sweep, "process_sweep": proc_sweep, "frame": f"{W}x{H} PNG, {size / 1e6:.2f} MB on disk (low-pass noise + fine noise)",
                                   "decoder": "Pillow in the loader threads (GIL released while decoding), frames written straight into the "
                                              "clip's pinned staging buffer (the `alloc` protocol of dataset.extract_dataset_clips)",
                                   "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
                                   "note": "PNG decode is outside the metric on both sides (SURVEY 8(d)); this is what a from-files run of the "
                                           "dataset driver sustains - never `value`"}
    return rec


def dataset_mode(args, eng, rank, world, barrier, precision):
    rec = dataset_pass(eng, args.workload, args.dataset_clips, args.clips_per_step, rank, world, args.gemm_split_k,
                       host_clips=args.host_clips, prefetch=args.prefetch, workers=args.loader_workers,
                       n_resident=args.resident_clips or 4, warmup=args.warmup, dump=args.dump_matrix,
                       frame_files=args.from_frame_files, loader_pools=getattr(args, "loader_pools", ()),
                       workers_sweep=[int(x) for x in args.loader_workers_sweep.split(",")] if args.loader_workers_sweep else None)
    if rank == 0:
        print(json.dumps(rec))
    if world > 1:
        rdist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the per-view acquisition path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* simulates ONE dataset: `--views-total` views (default 8: the "512^3 volume x 8 views" workload of
BASELINE.json, configs[1] per view, configs[2] sharding) of one ground-truth volume -- rotate -> attenuate -> PSF
convolve -> adjust -> slice extraction -> Poisson, device-resident (ground truth and all acquisitions stay in HBM).
With N > 1 ranks (one process per GPU) the views shard round-robin, view v -> rank v % N
(SimulateMultiViewDataset.java:567 iterates independent views), and every step contains one broadcast of a ground
truth from rank 0 over xGMI -- the only collective on the path, issued through the C ABI
(mvsim_comm_broadcast_volume: scatter + all-gather over all links).  Scaling is STRONG by default: the dataset has 8
views whatever N is (N = 8: one view per GPU).  The broadcast is issued one dataset ahead into the second of two
ground-truth buffers on its own HIP stream, so it overlaps the views of the current dataset (`--serial-broadcast` puts it
in front of them instead).  `--scaling weak` keeps 8 views PER GPU instead (dataset of 8 N views).

Launching: `python bench.py --gpus N ...` with N > 1 and no WORLD_SIZE in the environment starts its N ranks ITSELF (one
child process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, before this process has
imported torch or touched a GPU; a failing child ends the run with a non-zero status).  Under torchrun /
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the environment already names the ranks and this
process is one of them.  `--gpus` that disagrees with WORLD_SIZE is an error.

`value` is measured with the library's defaults (round 3: extract + Poisson of view v beside rotate+attenuate of view v+1,
PSF spectrum beside passes A/B -- bit-identical to the serial order); the `roofline` object comes from a SERIAL leg of the
same K steps (both overlaps off: one kernel at a time, so HIP-event stage times are the kernels' own durations and agree
with a rocprofv3 --kernel-trace of `bench.py --serial`).

Rank 0 prints ONE JSON line (schema in the task contract) including
  roofline     -- dominant stage: algorithmic bytes / HIP-event time against the 8 TB/s HBM peak, `fused_bytes` /
                  `frac_fused` on the bytes the fused path must move, plus the PMC-measured traffic of the same stage
                  (profiles/r06_traffic.json, valid only for the kernel sources it was measured on); with
                  --conv-method 2 the direct stencil against the 157.3 Tflop/s fp32 vector peak (bound "fp32")
  cpu_baseline -- the CPU oracle (C restatement of the reference's ImgLib2 path, not the JVM) on a bounded sample, two modes:
                  as_reference (the reference's threading) and all_cores
  value_dense  -- top level, beside `value`: the same workload without a single empty voxel (= no_empty_space.value)
  no_empty_space -- N = 1: the serial leg again on the phantom + 1e-6 (nothing for the exact zero-row fast paths to skip)
  poisson_queue -- the sampler's work queue after the timed steps: GiB held, what a view queued, its fullest block; `auto_share`: the same steps
                  with the queue sized from what the views need (option poisson_queue_share=auto): GiB and Mvoxel/s
  end_to_end   -- N = 1: the same views with page-locked HOST buffers in and out (PCIe-inclusive; never `value`); acquisitions cross as
                  uint16 counts, the float32 transfer is timed beside it
  size_1024    -- N = 1: one 1024^3 view, same stage timings and roofline keys, its own PMC traffic record.  N > 1: `--views-total` 1024^3
                  views sharded v % N with one 4.3 GB broadcast per step and the `multi_gpu` diagnostics of the main line
  tiled_1024   -- N > 1 (and --rehearse-multi): BASELINE configs[3] as stated -- 1024^3, 31 x 31 x 63 PSF, inc 4, six views, EVERY view cut into
                  N z slabs, one per rank (multiview-simulation_amd/tiling.py), the one double of adjustImage's sum reduced by the C ABI's
                  mvsim_comm_allreduce_sum_f64; per-rank slab / all-reduce / finish times and the halo-recompute share
  small_views  -- N = 1: the sizes the reference itself runs (128^3 x 8, 289^3 / 51^3 / inc 3 x 7, 256^3 x 8): sequential views against ONE
                  mvsim_simulate_views_dev call (views stacked), bit-identity checked
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP32_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak (same guide); the direct stencil's bound
# PMC traffic record of the running build (tools/profile_all.sh writes it first and points the bench at it through the environment)
TRAFFIC_JSON = os.environ.get("MVSIM_TRAFFIC_JSON") or os.path.join(ROOT, "profiles", "r06_traffic.json")
TRAFFIC_JSON_1024 = os.environ.get("MVSIM_TRAFFIC_JSON_1024") or os.path.join(ROOT, "profiles", "r06_traffic_1024.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512, help="cubic volume edge (512 = BASELINE configs[1])")
    ap.add_argument("--psf", type=int, default=31, help="cubic PSF edge")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: --views-total views per dataset whatever N is (BASELINE configs[2]: 8 views, one per GPU "
                         "at N = 8); weak: --views-per-gpu views on every GPU")
    ap.add_argument("--views-total", type=int, default=8)
    ap.add_argument("--views-per-gpu", type=int, default=8, help="weak scaling only")
    ap.add_argument("--inc", type=int, default=1, help="lightsheet spacing (1 = convolve+noise target)")
    ap.add_argument("--snr", type=float, default=25.0)
    ap.add_argument("--conv-method", type=int, default=1, help="1 FFT, 2 direct stencil")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer (PCIe-inclusive) record")
    ap.add_argument("--rehearse-multi", action="store_true",
                    help="N = 1 only: run the data path of N > 1 on the one GPU -- two ground-truth buffers, torch-owned view and "
                         "broadcast streams, the C ABI's RCCL communicator with one rank, one mvsim_comm_broadcast_volume per step "
                         "issued one dataset ahead (a rehearsal of the control flow; the line it prints is not a result)")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the two_streams sub-record (N = 1 only)")
    ap.add_argument("--serial", action="store_true",
                    help="switch the library's default overlaps (tail_overlap, psf_overlap) OFF for the main line: one kernel at a "
                         "time, what a rocprofv3 --kernel-trace profile should be taken with (per-kernel durations then add up to "
                         "the stage times); without it `value` uses the defaults and `roofline` comes from an extra serial leg")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launcher / rendezvous check only: every rank joins the process group, all-reduces a 1 and rank 0 prints "
                         "{n_gpus, ranks_seen}; no GPU, no libmvsim (what the CPU test of the self-launcher runs with --backend gloo)")
    ap.add_argument("--no-size-1024", action="store_true", help="skip the 1024^3 sub-record")
    ap.add_argument("--no-small-views", action="store_true",
                    help="skip the `small_views` sub-record (N = 1 only): the reference's own small sizes, views stacked in one call")
    ap.add_argument("--no-tiled-1024", action="store_true",
                    help="N > 1 data path only: skip the `tiled_1024` sub-record (BASELINE configs[3]: every 1024^3 view cut into N z slabs)")
    ap.add_argument("--no-dense-leg", action="store_true", help="skip the `no_empty_space` sub-record (N = 1 only)")
    ap.add_argument("--no-compact-queue-leg", action="store_true", help="skip `poisson_queue.auto_share` (N = 1 only)")
    ap.add_argument("--no-main-iteration", action="store_true",
                    help="skip the `main_iteration` sub-record (N = 1 only): whole iterations of the reference's view loop, device-resident")
    ap.add_argument("--cpu-slab", type=int, default=64, help="z extent of the CPU-baseline sample slab")
    ap.add_argument("--cpu-poisson-planes", type=int, default=64,
                    help="planes of the slab the reference-exact (inter-arrival) Poisson sampler is timed on; the rest is scaled")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts/HIP streams per GPU; with 2 the views alternate between them so that the VALU-bound "
                         "Poisson kernel of one view overlaps the HBM-bound passes of the next (per-kernel durations then "
                         "include time sharing; the default keeps the roofline clean)")
    ap.add_argument("--backend", default="auto",
                    help="torch.distributed backend for N > 1.  auto = gloo: torch.distributed is the CONTROL plane only (rendezvous, the "
                         "128-byte RCCL id, barriers, timing reductions -- host tensors), the data travels through the C ABI's own RCCL "
                         "communicator, so a rank holds ONE RCCL instance (VERDICT r5 weak #8).  nccl = torch's process group on RCCL as "
                         "well (rounds 1-5)")
    ap.add_argument("--collective", choices=("auto", "mvsim", "torch"), default="auto",
                    help="who broadcasts the ground truth: the C ABI's RCCL collective (auto: whenever every rank has a GPU of its own) or "
                         "torch.distributed (auto: ranks that share a GPU -- rehearsals; RCCL cannot place two ranks on one device)")
    ap.add_argument("--no-broadcast-ab", action="store_true",
                    help="N > 1 data path: time only --broadcast's form.  Default: after the line's own K steps the other forms of the "
                         "ground-truth broadcast (scatter_allgather, pipelined, peer_copy) run the same K steps back to back, every form's "
                         "numbers go to multi_gpu.broadcast_ab and the fastest one is reported as `value`")
    ap.add_argument("--broadcast", choices=("scatter_allgather", "ring", "peer_copy", "pipelined"), default="scatter_allgather",
                    help="form of the ground-truth broadcast in the C ABI: RCCL scatter + all-gather over all links (default), one RCCL "
                         "ring broadcast, the same scatter + all-gather as copy-engine transfers between IPC-mapped buffers "
                         "(peer_copy: no CUs taken from the views; two 16-byte all-reduces per broadcast remain as barriers), or the "
                         "pipelined form (the root scatters the whole volume as N - 1 chunks while the peers all-gather the pieces "
                         "that have arrived: ~S / ((N - 1) b) instead of 2 S / (N b); its schedule is checked on the CPU, the form has "
                         "never run with more than one rank)")
    ap.add_argument("--serial-broadcast", action="store_true",
                    help="N > 1: broadcast the ground truth at the start of each step instead of one step ahead")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here.  Called before torch is imported and before
    anything touches a GPU (the parent only waits); children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.  Rank 0's JSON
    line reaches this process's stdout because the children inherit it.  Any child that fails ends the others (by PID)
    and the parent exits with that child's status."""
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MVSIM_BENCH_SELF_LAUNCHED="1",
                   MVSIM_BENCH_LAUNCHER_PID=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))

    # SIGTERM / SIGINT to the launcher must not orphan its ranks (they would keep the GPUs and the rendezvous port): turn
    # the signal into an exception so that the `finally` below runs.  A SIGKILL cannot be caught here: for that case every
    # rank asks the kernel for a SIGTERM when its parent dies (die_with_parent, first thing in the child).
    def on_signal(signum, _frame):
        raise KeyboardInterrupt(f"signal {signum}")
    import signal
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write(f"bench.py: rank {r} exited with status {code}; stopping the other ranks\n")
                    for q in alive:
                        procs[q].terminate()
            if alive:
                time.sleep(0.05)
    finally:
        # interrupted (Ctrl-C, a caller's timeout): the ranks are OUR children -- end exactly those, by PID
        for q in alive:
            if procs[q].poll() is None:
                procs[q].terminate()
        for q in alive:
            try:
                procs[q].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[q].kill()
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


def die_with_parent() -> None:
    """A rank started by self_launch: have the kernel send SIGTERM when the launcher dies, however it dies (SIGKILL from a
    caller's timeout included).  prctl(PR_SET_PDEATHSIG) through ctypes, before torch or HIP are touched; if the launcher
    is gone already -- this process's parent is no longer the PID the launcher exported (a launcher that IS pid 1, a container's
    entry point, is a live parent; an orphan adopted by a subreaper has a parent that is neither 1 nor the launcher) -- leave now."""
    import signal
    try:
        libc = C.CDLL(None, use_errno=True)
        PR_SET_PDEATHSIG = 1
        libc.prctl(PR_SET_PDEATHSIG, int(signal.SIGTERM), 0, 0, 0)
    except Exception:
        return
    launcher = os.environ.get("MVSIM_BENCH_LAUNCHER_PID", "")
    if launcher.isdigit() and os.getppid() != int(launcher):
        sys.exit(1)


def rccl_report(mvs, torch) -> dict:
    """Which RCCL each side of this process is bound to: PyTorch ships its own librccl.so and loads it first; libmvsim.so
    names librccl.so.1 and gets whatever copy the loader already holds under that name."""
    mapped = []
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "librccl" in line:
                    path = line.split()[-1]
                    if path not in mapped:
                        mapped.append(path)
    except OSError:
        pass
    rep = {"mapped": mapped}
    try:
        rep["libmvsim"] = mvs.Context.comm_library_info()
    except Exception as e:
        rep["libmvsim"] = {"failed": repr(e)}
    try:
        v = torch.cuda.nccl.version()
        rep["torch"] = {"version": ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)}
    except Exception as e:
        rep["torch"] = {"failed": repr(e)}
    return rep


def cpu_baseline(gt: np.ndarray, psf_raw: np.ndarray, degrees: int, inc: int, snr: float, slab: int, poisson_planes: int) -> dict:
    """Time the CPU oracle -- a C restatement of the reference's ImgLib2 path, NOT the JVM (no JDK in the image) -- on a
    bounded sample of the same workload: a centred z-slab of the volume through all five stages, in two modes.
      as_reference: the reference's own threading -- everything a single-threaded cursor loop except the FFT convolution
                    (SimulateMultiViewDataset.java:257,527) -- and the reference-exact inter-arrival Poisson sampler on
                    java.util.Random (uncommons/PoissonGenerator.java:95-109), timed on `poisson_planes` planes of the slab
      all_cores:    OpenMP over planes / columns in rotate, attenuate, adjust; the counter-based sampler (the GPU's own
                    specification) over all cores; the same scipy float32 FFT convolution
    `value` is mode as_reference (what a user of the reference gets on this host)."""
    import oracle
    nz = gt.shape[0]
    slab = min(slab, nz)
    z0 = (nz - slab) // 2
    sub = np.ascontiguousarray(gt[z0:z0 + slab])
    vox = sub.size
    n_extract = (slab - 1) // inc + 1
    cores = os.cpu_count()
    delta = float(np.float32(0.01))

    def stages(parallel: bool):
        oracle.set_parallel(parallel)
        try:
            t = {}
            t0 = time.perf_counter(); rot = oracle.rotate_around_axis(sub, 0, degrees); t["rotate"] = time.perf_counter() - t0
            t0 = time.perf_counter(); att = oracle.attenuate3d(rot, delta); t["attenuate"] = time.perf_counter() - t0
            psf = psf_raw.copy()
            t0 = time.perf_counter(); con = oracle.convolve_fft(att, psf, workers=-1); t["convolve_fft"] = time.perf_counter() - t0
            t0 = time.perf_counter(); oracle.adjust_image(con, 1e-4, 1.0); t["adjust"] = time.perf_counter() - t0
            return t, con
        finally:
            oracle.set_parallel(False)

    modes = {}
    t, con = stages(False)
    nsl = max(1, min(poisson_planes, n_extract))
    t0 = time.perf_counter()
    oracle.extract_slices_ref(con[: (nsl - 1) * inc + 1], inc, snr, oracle.JRandom(464232194))
    measured = time.perf_counter() - t0
    t["extract_poisson"] = measured * (n_extract / nsl)
    total = sum(t.values())
    modes["as_reference"] = {
        "value": vox / total / 1e6, "unit": "Mvoxel/s",
        "threads": {"convolve_fft": cores, "rotate": 1, "attenuate": 1, "adjust": 1, "extract_poisson": 1},
        "sampler": "reference-exact inter-arrival sampler on java.util.Random",
        "seconds": {k: round(v, 3) for k, v in t.items()},
        "poisson_planes_timed": nsl, "poisson_planes_total": n_extract,
        "extrapolated_fraction_of_seconds": round((t["extract_poisson"] - measured) / total, 4),
    }
    t2, con2 = stages(True)
    t0 = time.perf_counter()
    oracle.extract_slices_counter(con2, inc, snr, 464232194, 0)
    t2["extract_poisson"] = time.perf_counter() - t0
    total2 = sum(t2.values())
    omp = oracle.max_threads()
    modes["all_cores"] = {
        "value": vox / total2 / 1e6, "unit": "Mvoxel/s",
        "threads": {"convolve_fft": cores, "rotate": omp, "attenuate": omp, "adjust": omp, "extract_poisson": omp},
        "sampler": "counter-based sampler (Philox + inversion / PTRS: the GPU path's own specification)",
        "seconds": {k: round(v, 3) for k, v in t2.items()},
        "extrapolated_fraction_of_seconds": 0.0,
    }
    return {
        "value": modes["as_reference"]["value"], "unit": "Mvoxel/s", "cores": cores, "kind": "port",
        "what": "C restatement of the reference's ImgLib2 path (oracle/), not the JVM; value = mode as_reference",
        "sample": (f"{sub.shape[2]}x{sub.shape[1]}x{slab} z-slab of the same view, {psf_raw.shape[0]}^3 PSF, all five stages; "
                   f"scipy float32 FFT convolution on all cores in both modes; reference-exact Poisson timed on {nsl} of "
                   f"{n_extract} planes"),
        "modes": modes,
    }


def load_traffic(n: int, psf: int, inc: int, streams: int, conv_method: int, kernel_sha: str, path: str = None):
    """PMC-measured HBM bytes per view and stage (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes), written
    by tools/pmc_traffic.py --json.  Valid only for the workload and the kernel sources it was collected on."""
    path = path or TRAFFIC_JSON
    if not os.path.exists(path):
        return None, f"{os.path.relpath(path, ROOT)} not present"
    rec = json.load(open(path))
    rec["file"] = os.path.relpath(path, ROOT)
    w = rec.get("workload", {})
    if (w.get("size"), w.get("psf"), w.get("inc")) != (n, psf, inc):
        return None, f"profiled workload {w} differs from this run"
    if streams != 1 or conv_method != 1:
        return None, "profiled with one stream on the FFT path; this run differs"
    if rec.get("kernel_sha") != kernel_sha:
        return None, f"kernel sources changed since the counters were collected ({rec.get('kernel_sha')} != {kernel_sha})"
    return rec, None


def roofline_record(mvs, stage: dict, nvox: int, nprime: int, n: int, psf_edge: int, conv_method: int, traffic, traffic_note,
                    view_wall_ms: float | None = None, overlapped: bool = False, plane_stats=None):
    """Roofline object from per-stage HIP-event times (ms).  Two byte models, never blended:
      algorithmic_bytes / frac       -- SURVEY.md 8(d): every REFERENCE stage reads its input once and writes its output once
                                        (rotate 8N + attenuate 8N; convolve 8N + 4K^3; extract + Poisson 8N'; view 24N + 8N')
      fused_bytes / frac_fused       -- what the fused path must move: rotate+attenuate is ONE kernel whose `rot` never
                                        exists in HBM (8N), so a view is 16N + 8N' (+ 4K^3); this fraction cannot exceed 1
    With conv_method 2 the convolve stage is the direct stencil: 2 K^3 N flop against the fp32 vector peak."""
    k3 = psf_edge ** 3
    alg = {
        "rotate_attenuate": 16 * nvox,               # rotate 8N + attenuate 8N (two reference stages)
        "convolve": 8 * nvox + 4 * k3,               # + PSF
        "extract_poisson": 8 * nprime,
    }
    fused = {"rotate_attenuate": 8 * nvox, "convolve": 8 * nvox + 4 * k3, "extract_poisson": 8 * nprime}
    ms = {
        "rotate_attenuate": stage["rotate_ms"] + stage["attenuate_ms"],
        "convolve": stage["psf_ms"] + stage["convolve_ms"] + stage["adjust_ms"],
        "extract_poisson": stage["extract_ms"],
    }
    per_view = (traffic or {}).get("per_view_bytes", {})
    stages = {}
    for k in alg:
        gbps = alg[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0
        fg = fused[k] / (ms[k] * 1e-3) / 1e9 if ms[k] > 0 else 0.0
        stages[k] = {"algorithmic_bytes": alg[k], "ms": round(ms[k], 4), "GBps": gbps, "frac": gbps / HBM_PEAK_GBS,
                     "fused_bytes": fused[k], "frac_fused": fg / HBM_PEAK_GBS}
        if k in per_view and ms[k] > 0:
            stages[k]["traffic"] = per_view[k]
            stages[k]["hbm_measured"] = per_view[k] / (ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS
    # the x transform of the convolution (pass A) runs INSIDE the rotate + attenuate kernel when the fused kernel is active
    # (k_rotate_attenuate_fftx): the convolve stage then holds the PSF spectrum and passes B..E only -- said in so many words
    # below, and the two stages are also reported as one (`rotate_attenuate_convolve`) so that nothing hides in the split
    x_in_rotate = conv_method == 1 and stage.get("pass_a_ms", 0.0) == 0.0 and stage.get("pass_b_ms", 0.0) > 0.0
    rc_ms = ms["rotate_attenuate"] + ms["convolve"]
    rc_alg, rc_fused = alg["rotate_attenuate"] + alg["convolve"], (8 * nvox + 4 * k3 if x_in_rotate else fused["rotate_attenuate"] + fused["convolve"])
    stages["rotate_attenuate_convolve"] = {
        "algorithmic_bytes": rc_alg, "ms": round(rc_ms, 4), "GBps": rc_alg / (rc_ms * 1e-3) / 1e9 if rc_ms > 0 else 0.0,
        "frac": rc_alg / (rc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rc_ms > 0 else 0.0,
        "fused_bytes": rc_fused, "frac_fused": rc_fused / (rc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rc_ms > 0 else 0.0,
        "note": "stages rotate_attenuate + convolve taken together (24 N + 4 K^3 algorithmic bytes; fused: ground truth in, convolved volume out)"}
    dom = max(ms, key=ms.get)
    b_view = 24 * nvox + 8 * nprime
    b_view_fused = (8 * nvox + 8 * nprime + 4 * k3) if x_in_rotate else (16 * nvox + 8 * nprime + 4 * k3)
    b_cn = 8 * nvox + 8 * nprime
    cn_ms = ms["convolve"] + ms["extract_poisson"]
    names = {
        "convolve": "convolve stage: PSF (x,y) spectrum (k_fft_x_r2c, k_fft_lines<FWD,sparse>), k_fft_x_r2c, k_fft_lines<FWD>, "
                    "k_zconv (direct z convolution), k_fft_lines<INV>, k_fft_x_c2r (+ adjust/Poisson epilogue when fused), "
                    "k_reduce_partials",
        "extract_poisson": "extract stage: k_extract4_noise2 + k_poisson_resolve",
        "rotate_attenuate": "k_rotate_attenuate_fftx (rotate + attenuate + x transform)" if x_in_rotate else "k_rotate_attenuate_axis0_lds",
    }
    if x_in_rotate:
        names["convolve"] = ("convolve stage WITHOUT its x transform (pass A runs inside k_rotate_attenuate_fftx, stage rotate_attenuate): PSF "
                             "(x,y) spectrum (k_fft_x_r2c, k_fft_lines<FWD,sparse>), k_fft_lines<FWD>, k_zconv, k_fft_lines<INV>, k_fft_x_c2r, "
                             "k_reduce_partials")
    passes = None
    geo = (C.c_int64 * 5)()
    if conv_method == 1 and mvs._lib.load().mvsim_fft_geometry((C.c_int64 * 3)(n, n, n), (C.c_int64 * 3)(psf_edge, psf_edge, psf_edge), geo) == 0:
        px, py_, planes, hxp, zdirect = (int(v) for v in geo)
        cplx = 8 * hxp * py_ * planes
        # planes the passes really processed: a specimen in empty space leaves planes whose spectrum is exactly zero, and the passes
        # skip them -- the byte model follows (mvsim_get_plane_stats of the last flagged view), so that no pass shows a
        # rate it did not run at.  fin: share of the planes that enter pass B / the z pass, fout: share that leaves the z pass.
        fin = fout = 1.0
        if plane_stats and plane_stats[0] == planes and zdirect:
            fin = 1.0 - plane_stats[1] / planes
            fout = 1.0 - plane_stats[2] / planes
        pb = {"A k_fft_x_r2c": (4 * nvox + cplx, stage["pass_a_ms"]),
              "B k_fft_lines<FWD>": (2 * cplx * fin, stage["pass_b_ms"]),
              ("C k_zconv" if zdirect else "C k_fft_lines<CONV>"): (cplx * fin + cplx * fout + (0 if zdirect else cplx), stage["pass_c_ms"]),
              "D k_fft_lines<INV>": (2 * cplx * fout, stage["pass_d_ms"]),
              "E k_fft_x_c2r": (cplx * fout + 4 * nvox, stage["pass_e_ms"])}
        passes = {k: {"bytes": int(b), "ms": round(t, 4), "GBps": b / (t * 1e-3) / 1e9, "frac": b / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
                  for k, (b, t) in pb.items() if t > 0}
        passes["planes"] = {"of_the_spectrum": planes, "share_entering_the_y_and_z_passes": round(fin, 4), "share_leaving_the_z_pass": round(fout, 4),
                            "note": "bytes = the pass's own reads + writes of the planes it processed (empty planes are skipped exactly, "
                                    "option skip_empty); a dense volume has both shares at 1"}
    view_ms = view_wall_ms if view_wall_ms else stage["total_ms"]
    if conv_method == 2:
        flop = 2.0 * k3 * nvox
        tf = flop / (ms["convolve"] * 1e-3) / 1e12 if ms["convolve"] > 0 else 0.0
        sg = (C.c_int64 * 5)()
        chunk = None
        if mvs._lib.load().mvsim_stencil_geometry((C.c_int64 * 3)(psf_edge, psf_edge, psf_edge), sg) == 0:
            chunk = {"psf_chunk_taps": [int(sg[0]), int(sg[1]), int(sg[2])], "lds_bytes_per_block": int(sg[3]), "blocks_per_cu": int(sg[4])}
        head = {"bound": "fp32", "kernel": "k_stencil_pair (LDS-tiled direct 3-D convolution, packed fp32 FMAs)",
                "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "traffic": None,
                "flop": flop, "flop_model": "2 * K^3 * N per view (useful taps only; the zero taps that pad a PSF row chunk are not counted)",
                "launch_ms": ms["convolve"], "stencil_geometry": chunk}
    else:
        head = {"bound": "hbm", "kernel": names[dom],
                "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": stages[dom]["frac"],
                "fused_bytes": fused[dom], "frac_fused": stages[dom]["frac_fused"],
                # HBM bytes of the dominant stage per view from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE,
                # separate passes of `bench.py --serial` (profiles/r06_traffic.json); null when that record does not describe
                # this build / workload
                "traffic": stages[dom].get("traffic"),
                "hbm_measured": stages[dom].get("hbm_measured"),
                "algorithmic_bytes": alg[dom], "launch_ms": ms[dom],
                "x_transform_in_rotate_kernel": x_in_rotate}
    rec = dict(head)
    rec.update({
        "stages": stages,
        # wall clock per view when the caller has it (with the overlaps on, the stage events of consecutive views overlap, so
        # their sum, total_ms, is no longer the time a view takes)
        "whole_view": {"bytes": b_view, "ms": view_ms, "frac": b_view / (view_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "fused_bytes": b_view_fused, "frac_fused": b_view_fused / (view_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "convolve_noise": {"bytes": b_cn, "ms": cn_ms, "frac": b_cn / (cn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           **({"note": "the convolution's x transform runs inside the rotate + attenuate kernel and is NOT in `ms`; "
                                       "`frac_with_rotate_kernel` charges that whole kernel (rotation and attenuation included) to the sub-path",
                               "frac_with_rotate_kernel": b_cn / ((cn_ms + ms["rotate_attenuate"]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
                              if x_in_rotate else {})},
        "stage_ms": {k: round(v, 4) for k, v in stage.items()},
        "passes": passes,
        "timed": ("overlapped views (library defaults): the stage times of rotate+attenuate and extract+Poisson include each other's "
                  "share of the chip" if overlapped else
                  "serial leg: tail_overlap = psf_overlap = 0, one kernel at a time on one stream (HIP events on that stream)"),
    })
    if traffic is None:
        rec["traffic_note"] = traffic_note
    else:
        rec["traffic_source"] = {"file": traffic.get("file"), "kernel_sha": traffic.get("kernel_sha"),
                                 "views_profiled": traffic.get("views_profiled")}
        tv = sum(per_view.values())
        rec["whole_view"]["traffic"] = tv
        rec["whole_view"]["hbm_measured"] = tv / (view_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return rec


def end_to_end_record(mvs, dev_index: int, gt_host: np.ndarray, psfs_raw: list, angles: list, inc: int, snr: float) -> dict:
    """The JNI boundary's view of the same workload: page-locked HOST buffers in and out through
    mvsim_simulate_view_async / mvsim_wait (upload(v+1) || compute(v) || download(v-1)).  Two flavours: the view loop of
    `main` (the SAME ground truth for every angle, SMVD:567-585: uploaded once, one acquisition comes back per view) and
    a fresh ground truth for every view (0.54 GB up + 0.54 GB down per 512^3 view)."""
    n = gt_host.shape[0]
    nzo = (n - 1) // inc + 1
    out = {}
    with mvs.Context(dev_index) as c:
        gts = [c.pinned_empty(gt_host.shape) for _ in range(2)]
        for g in gts:
            g[...] = gt_host
        acq = [c.pinned_empty((nzo, n, n)) for _ in range(3)]
        params = [c.view_params(degrees=a, inc=inc, snr=snr, seed=464232194, stream=v, conv_method=1) for v, a in enumerate(angles)]

        def run(fresh_gt: bool, reps: int):
            tickets = []
            gen = 0
            t0 = time.perf_counter()
            for r in range(reps):
                for v in range(len(angles)):
                    i = r * len(angles) + v
                    if fresh_gt:
                        gen += 1
                    tickets.append(c.simulate_view_async(gts[i % 2] if fresh_gt else gts[0], psfs_raw[v % len(psfs_raw)].copy(), params[v],
                                                         {"acq": acq[i % 3]}, gt_generation=gen))
                    if i >= 2:
                        c.wait(tickets[i - 2])                      # keeps at most two views outstanding: acq[i % 3] is free again
            for t in tickets[-2:]:
                c.wait(t)
            return (time.perf_counter() - t0) / (reps * len(angles))
        run(False, 1)                                               # warm-up: workspaces, page tables
        for key, fresh in (("same_ground_truth", False), ("fresh_ground_truth_per_view", True)):
            dt = run(fresh, 2)
            out[key] = {"ms_per_view": dt * 1e3, "views_per_s": 1.0 / dt, "Mvoxel_per_s": n ** 3 / dt / 1e6,
                        "host_bytes_per_view": (4 * n ** 3 if fresh else 0) + 2 * n * n * nzo}
        assert float(acq[0].max()) > 0
        u16_views, u16_fallbacks = c.transfer_stats()
        out["acquisition_transfer"] = {"views_as_uint16": u16_views, "fell_back_to_float32": u16_fallbacks,
                                       "note": "Poisson counts (Tools.java:84 stores them as floats) cross PCIe as uint16 and are widened into the caller's "
                                               "float32 buffer by mvsim_wait (host threads, streaming stores); a view with a count beyond 65 535 is fetched "
                                               "as float32 automatically; identical arrays either way"}
        c.set_option("acq_transfer", "f32")
        dt = run(False, 2)
        out["same_ground_truth_float32_transfer"] = {"ms_per_view": dt * 1e3, "Mvoxel_per_s": n ** 3 / dt / 1e6, "host_bytes_per_view": 4 * n * n * nzo}
        del gts, acq
    out["note"] = ("page-locked host buffers in and out (mvsim_host_alloc), mvsim_simulate_view_async + mvsim_wait: two staging "
                   "sets, three HIP streams; PCIe-inclusive, reported beside `value`, never as `value`")
    return out


def size_1024_record(mvs, torch, dev, dev_index: int, gt_dev_512, psf_raw: np.ndarray, inc: int, snr: float) -> dict:
    """north_star's second size: one 1024^3 view (31^3 PSF, inc as the main run), device-resident.  `value` with the library
    defaults (three back-to-back views), stage times by HIP events from a serial leg.  The ground truth is the 512^3
    phantom up-sampled 2x on the device (spheres of twice the radius: the character of the phantom at that size,
    SimulateMultiViewDataset.java:436-522 scales the radii with the canvas)."""
    n = 1024
    g = gt_dev_512.view(512, 512, 512)
    g = g.repeat_interleave(2, dim=0).repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous().view(-1)
    nzo = (n - 1) // inc + 1
    acq = torch.empty(n * n * nzo, dtype=torch.float32, device=dev)
    with mvs.Context(dev_index) as c:
        p = c.view_params(degrees=60, inc=inc, snr=snr, seed=464232194, stream=0, conv_method=1)
        c.simulate_view_dev(g.data_ptr(), (n, n, n), psf_raw.copy(), p, acq.data_ptr())
        c.synchronize()
        reps = 3

        def run():
            t0 = time.perf_counter()
            for _ in range(reps):
                c.simulate_view_dev(g.data_ptr(), (n, n, n), psf_raw.copy(), p, acq.data_ptr())
            c.synchronize()
            return (time.perf_counter() - t0) / reps
        wall = run()
        c.set_option("tail_overlap", 0)
        c.set_option("psf_overlap", 0)
        c.enable_timing(True)
        wall_serial = run()
        stage = c.timings()
        c.enable_timing(False)
        build = importlib.import_module("multiview-simulation_amd.build")
        traffic, tnote = load_traffic(n, psf_raw.shape[0], inc, 1, 1, build.source_sha(), TRAFFIC_JSON_1024)
        rl = roofline_record(mvs, stage, n ** 3, n * n * nzo, n, psf_raw.shape[0], 1, traffic, tnote,
                             view_wall_ms=wall_serial * 1e3, plane_stats=c.plane_stats())
    mean_count = float(acq[: n * n].double().mean().item())
    del acq, g
    torch.cuda.empty_cache()
    return {"workload": f"1024^3 float volume, 1 view, {psf_raw.shape[0]}^3 PSF, inc={inc}, SNR {snr:g}, device-resident",
            "views": reps, "ms_per_view": wall * 1e3, "value": n ** 3 / wall / 1e6, "unit": "Mvoxel/s",
            "serial": {"ms_per_view": wall_serial * 1e3, "value": n ** 3 / wall_serial / 1e6},
            "first_plane_mean_count": mean_count, "roofline": rl}


def small_views_record(mvs, synth, dev_index: int) -> dict:
    """Views that cannot fill the chip one at a time -- the sizes the REFERENCE itself runs: BASELINE configs[0] (128^3, 15^3 PSF) and
    the run `main` ships with (289^3 phantom, 51^3 PSF stacks, lightsheet spacing 3, seven views: SimulateMultiViewDataset.java:376-380,
    399, 531-548, loop :567), plus 256^3.  Per size: V sequential mvsim_simulate_view_dev calls against ONE mvsim_simulate_views_dev call
    (the views stacked: one launch per stage for all of them), wall clock over `reps` datasets, device-resident; the acquisitions of the
    two forms are compared bit for bit."""
    out = {}
    for name, n, k, inc, nv, reps in (("128^3_psf15_inc1_x8", 128, 15, 1, 8, 40), ("289^3_psf51_inc3_x7", 289, 51, 3, 7, 30), ("256^3_psf31_inc1_x8", 256, 31, 1, 8, 20)):
        gt = synth.sphere_phantom(n)
        nzo = (n - 1) // inc + 1
        psfs = [synth.gaussian_psf(k, sigma=(k / 15.0, k / 14.0, k / 5.0 + 0.05 * v)) for v in range(nv)]
        with mvs.Context(dev_index) as c:
            d_gt = c.dev_alloc(gt.nbytes)
            c.upload(d_gt, gt)
            acq = [c.dev_alloc(nzo * n * n * 4) for _ in range(nv)]
            params = [c.view_params(degrees=15 + (360 * v) // nv, inc=inc, snr=25.0, seed=464232194, stream=v, conv_method=1) for v in range(nv)]

            def sequential():
                for v in range(nv):
                    c.simulate_view_dev(d_gt, (n, n, n), psfs[v].copy(), params[v], acq[v])

            def stacked():
                c.simulate_views_dev(d_gt, (n, n, n), [p.copy() for p in psfs], params, acq)

            def clock(fn):
                for _ in range(3):
                    fn()
                c.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                c.synchronize()
                return (time.perf_counter() - t0) / (reps * nv) * 1e3
            t_seq = clock(sequential)
            want = [c.download(a, (nzo, n, n)) for a in acq]
            t_st = clock(stacked)
            same = all(np.array_equal(c.download(a, (nzo, n, n)), w) for a, w in zip(acq, want))
            for d in acq + [d_gt]:
                c.dev_free(d)
        out[name] = {"views": nv, "sequential_ms_per_view": round(t_seq, 4), "sequential_Mvoxel_per_s": n ** 3 / t_seq / 1e3,
                     "ms_per_view": round(t_st, 4), "value": n ** 3 / t_st / 1e3, "unit": "Mvoxel/s", "bit_identical_to_sequential": bool(same)}
    out["note"] = ("value = one mvsim_simulate_views_dev call per dataset (views stacked: the view index in the kernels' grids, one launch per "
                   "stage for all views); sequential = one mvsim_simulate_view_dev call per view; wall clock, device-resident, Python call overhead included")
    return out


class Env:
    """What every leg of a run shares: the modules, this rank's device, the process group's geometry and -- for the N > 1 data path --
    the broadcast stream and the context that holds the C ABI's own RCCL communicator."""


class ShardedRun:
    """The view loop of `main` (SimulateMultiViewDataset.java:567) sharded over the ranks: a STEP is one dataset of `total_views` views of
    one n^3 ground truth, view v on rank v % world.  With env.multi every step also contains one broadcast of a ground truth from rank 0
    (mvsim_comm_broadcast_volume on env.bc_ctx, or torch.distributed for the gloo rehearsals), issued one dataset ahead into the second of
    two buffers on env.bc_stream unless serial_broadcast.  Used for the main line (512^3) and for the `size_1024` leg of N > 1."""

    def __init__(self, env, n, kdim, sigma, inc, snr, conv_method, total_views, streams, serial, serial_broadcast, fill_gt):
        torch, mvs, synth = env.torch, env.mvs, env.synth
        self.env, self.n, self.inc, self.total_views = env, n, inc, total_views
        self.serial, self.serial_broadcast = serial, serial_broadcast
        self.dims, self.nvox = (n, n, n), n ** 3
        self.nzo = (n - 1) // inc + 1
        self.my_views = mvs.shard_views(total_views, env.world, env.rank)
        self.angles = [15 + (360 * v) // total_views for v in range(total_views)]      # 8 views: 45-degree steps (configs[2])
        # N > 1 keeps two ground-truth buffers so that the broadcast of the next dataset runs (RCCL, own stream) while the views of
        # the current one are being computed
        self.gt_bufs = [torch.empty(self.nvox, dtype=torch.float32, device=env.dev) for _ in range(2 if env.multi else 1)]
        if env.rank == 0:
            for b in self.gt_bufs:
                fill_gt(b)
        # one PSF per view (the reference loads Angle<k>.tif per view, SMVD:579): vary sigma_z slightly so no spectrum can be shared
        self.psfs = [synth.gaussian_psf(*kdim, sigma=(sigma[0], sigma[1], sigma[2] + 0.05 * (v % 8))) for v in self.my_views]
        self.acq = [torch.empty(n * n * self.nzo, dtype=torch.float32, device=env.dev) for _ in self.my_views]
        # one context (own HIP stream + workspaces) per concurrent view pipeline
        self.ctxs = [mvs.Context(env.dev_index) for _ in range(max(1, streams))]
        self.params = [self.ctxs[0].view_params(degrees=self.angles[v], inc=inc, snr=snr, seed=464232194, stream=v, conv_method=conv_method)
                       for v in self.my_views]
        self.set_overlap(not serial)
        self.view_streams = []
        self.registered = []
        if env.multi:
            # the view pipelines run on torch-owned HIP streams so that torch events can order them against the broadcast stream
            # without blocking the host
            self.view_streams = [torch.cuda.Stream(device=env.dev) for _ in self.ctxs]
            for c, vs in zip(self.ctxs, self.view_streams):
                c.set_stream(vs.cuda_stream)
            if env.bc_ctx is not None and env.broadcast == "peer_copy":
                # the copy-engine form writes into the peers' buffers through IPC mappings: an explicit, collective registration of
                # every buffer a broadcast will fill (never a cache keyed by address)
                for b in self.gt_bufs:
                    env.bc_ctx.comm_register_volume(b.data_ptr(), self.nvox)
                    self.registered.append(b.data_ptr())
        self.views_done = [[], []]      # per ground-truth buffer: events after the last views that read it
        self.bcast_done = [None, None]  # per ground-truth buffer: event after the broadcast that filled it
        self.step_no = 0
        self.bc_events = []             # (start, end) timing events around every broadcast of a timed region (on bc_stream)
        self.view_events = []           # per timed step: (starts, ends) around this rank's views, one pair per view stream
        self.record_diag = False

    def set_overlap(self, on: bool):
        for c in self.ctxs:
            # on a caller's stream (N > 1) the tail overlap needs the explicit opt-in ("any"): nothing here reads a view's output from
            # another stream before the final device-wide synchronisation, and the events that gate the next broadcast only protect
            # the ground truth, which the tail does not read
            c.set_option("tail_overlap", ("any" if self.env.multi else 1) if on else 0)
            c.set_option("psf_overlap", 1 if on else 0)

    def set_broadcast(self, form: str):
        """Another form of the ground-truth broadcast for the steps that follow (mvsim_set_option "broadcast" on the communicator's
        context); the copy-engine form needs its buffers registered first -- an explicit collective, every rank calls it."""
        env = self.env
        self.sync()
        if form == "peer_copy" and not self.registered:
            for b in self.gt_bufs:
                env.bc_ctx.comm_register_volume(b.data_ptr(), self.nvox)
                self.registered.append(b.data_ptr())
        env.bc_ctx.set_option("broadcast", form)
        env.broadcast = form

    def issue_broadcast(self, b):
        env, torch = self.env, self.env.torch
        with torch.cuda.stream(env.bc_stream):
            for e in self.views_done[b]:
                env.bc_stream.wait_event(e)             # readers of the previous contents have finished
            if self.record_diag:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(env.bc_stream)
            if env.bc_ctx is not None:
                env.bc_ctx.comm_broadcast_volume(self.gt_bufs[b].data_ptr(), self.nvox, 0)      # enqueued on bc_stream
            else:
                work = env.dist.broadcast(self.gt_bufs[b], src=0, async_op=True)
                work.wait()                             # nccl: bc_stream waits for the collective; gloo: host waits
            e = torch.cuda.Event(enable_timing=self.record_diag)
            e.record(env.bc_stream)
            self.bcast_done[b] = e
            if self.record_diag:
                self.bc_events.append((e0, e))

    def step(self):
        env, torch = self.env, self.env.torch
        cur = 0
        if env.multi:
            cur = self.step_no % 2
            if self.serial_broadcast or self.bcast_done[cur] is None:
                self.issue_broadcast(cur)               # this dataset's ground truth (prologue / serial mode)
            for vs in self.view_streams:
                vs.wait_event(self.bcast_done[cur])     # ground truth has landed before the views read it
            if not self.serial_broadcast:
                self.issue_broadcast(1 - cur)           # next dataset's ground truth, overlapped with these views
        gt_ptr = self.gt_bufs[cur].data_ptr()
        diag = env.multi and self.record_diag and self.my_views
        if diag:
            v0 = []
            for vs in self.view_streams:
                e = torch.cuda.Event(enable_timing=True)
                e.record(vs)
                v0.append(e)
        for i in range(len(self.my_views)):
            self.ctxs[i % len(self.ctxs)].simulate_view_dev(gt_ptr, self.dims, self.psfs[i].copy(), self.params[i], self.acq[i].data_ptr())
        if diag:
            v1 = []
            for c, vs in zip(self.ctxs, self.view_streams):
                c.join()                                # a pending tail belongs to this step's views
                e = torch.cuda.Event(enable_timing=True)
                e.record(vs)
                v1.append(e)
            self.view_events.append((v0, v1))
        if env.multi:
            self.views_done[cur] = []
            for vs in self.view_streams:
                e = torch.cuda.Event()
                e.record(vs)
                self.views_done[cur].append(e)
            self.step_no += 1

    def sync(self):
        self.env.torch.cuda.synchronize()
        if self.env.world > 1:
            self.env.dist.barrier()
        self.env.torch.cuda.synchronize()

    def check_broadcast(self):
        """Outside any timed region: every rank must hold rank 0's ground truth in each buffer a broadcast has filled."""
        torch, dist = self.env.torch, self.env.dist
        for b_i, done in enumerate(self.bcast_done):
            if done is None:
                continue
            chk = torch.stack([self.gt_bufs[b_i].double().sum(), self.gt_bufs[b_i].double().abs().max()])
            if self.env.backend != "nccl":
                chk = chk.cpu()
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if not torch.equal(lo, hi) or float(hi[1]) == 0.0:
                raise SystemExit(f"rank {self.env.rank}: ground-truth buffer {b_i} differs between ranks after the broadcast")
        self.sync()

    def timed(self, steps):
        """Exactly `steps` steps between two barrier + synchronise pairs.  Returns (seconds: MAX over ranks, multi_gpu diagnostics or None)."""
        env, torch, dist = self.env, self.env.torch, self.env.dist
        self.bc_events, self.view_events = [], []
        self.record_diag = env.multi
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.sync()
        own = time.perf_counter() - t0
        self.record_diag = False
        diag = None
        if env.multi:
            # What a scaling run needs to be attributable: per step, the broadcast's own duration on its stream, this rank's views on
            # theirs, how much of the broadcast the views hid, and every rank's own wall clock.  Events, read after the timed region;
            # nothing here is inside it but the event records themselves.
            bc_ms = [a.elapsed_time(b) for a, b in self.bc_events]
            vw_ms = [max(a.elapsed_time(b) for a, b in zip(v0, v1)) for v0, v1 in self.view_events]   # the busiest of this rank's view streams
            mine = torch.tensor([own / steps * 1e3, sum(bc_ms) / max(1, len(bc_ms)), sum(vw_ms) / max(1, len(vw_ms)),
                                 float(len(self.my_views))], dtype=torch.float64, device=env.dev if env.backend == "nccl" else "cpu")
            every = [torch.zeros_like(mine) for _ in range(env.world)]
            if env.world > 1:
                dist.all_gather(every, mine)
            else:
                every = [mine]
            rows = [[float(x) for x in t.tolist()] for t in every]
            step_ms = [r[0] for r in rows]
            b_ms = max(r[1] for r in rows)                 # the collective ends when its slowest rank does
            v_ms = max(r[2] for r in rows)
            exposed = min(max(max(step_ms) - v_ms, 0.0), b_ms) if b_ms > 0 else 0.0
            diag = {"broadcast_ms": round(b_ms, 4), "views_ms": round(v_ms, 4),
                    "broadcast_hidden_frac": round(1.0 - exposed / b_ms, 4) if b_ms > 0 else None,
                    "broadcast_GBps_per_rank": round(4 * self.nvox / (b_ms * 1e-3) / 1e9, 2) if b_ms > 0 else None,
                    "ms_per_step_min": round(min(step_ms), 4), "ms_per_step_max": round(max(step_ms), 4),
                    "per_rank": [{"rank": i, "ms_per_step": round(r[0], 4), "broadcast_ms": round(r[1], 4), "views_ms": round(r[2], 4),
                                  "views": int(r[3])} for i, r in enumerate(rows)],
                    "note": "per step: broadcast_ms = scatter + all-gather of the next dataset's ground truth on its own stream (HIP events, "
                            "slowest rank); views_ms = this dataset's views on the busiest rank; broadcast_hidden_frac = share of the broadcast "
                            "that did not extend the step beyond the views (1 = fully hidden); broadcast_GBps_per_rank = volume bytes / "
                            "broadcast_ms (what every rank received); expected from link rates: DESIGN.md section 6"}
        elapsed = own
        if env.world > 1:
            tt = torch.tensor([own], dtype=torch.float64, device=env.dev if env.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, diag

    def close(self):
        self.sync()
        for p in self.registered:
            self.env.bc_ctx.comm_unregister_volume(p)
        self.registered = []
        for c in self.ctxs:
            c.close()
        self.ctxs = []
        self.acq.clear()
        self.gt_bufs.clear()
        self.env.torch.cuda.empty_cache()


def upsample2x(torch, g512, n):
    """The 512^3 phantom up-sampled 2x on the device (spheres of twice the radius: the character of the phantom at that size,
    SimulateMultiViewDataset.java:436-522 scales the radii with the canvas)."""
    g = g512.view(n, n, n)
    return g.repeat_interleave(2, dim=0).repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous().view(-1)


def sharded_2x_leg(env, gt_dev, n_half, psf_edge, inc, snr, total_views, steps, serial_broadcast):
    """north_star's second size on the N > 1 data path (n_half = 512: `total_views` views of a 1024^3 ground truth sharded v % N, one
    4.3 GB broadcast per step; same machinery and diagnostics as the main line).  Returns (record, the ground truth every rank now holds)."""
    torch = env.torch
    n = 2 * n_half

    def fill(buf):
        buf.copy_(upsample2x(torch, gt_dev, n_half))
    run = ShardedRun(env, n, (psf_edge,) * 3, (2.0, 2.2, 6.0), inc, snr, 1, total_views, 1, serial=False, serial_broadcast=serial_broadcast,
                     fill_gt=fill)
    try:
        run.step()
        run.sync()
        if env.world > 1:
            run.check_broadcast()
        elapsed, diag = run.timed(steps)
        mean_count = float(run.acq[0][: n * n].double().mean().item()) if run.acq else None
        gt_keep = run.gt_bufs[0]                       # every rank holds the 1024^3 ground truth now: the tiled leg reuses it
        run.gt_bufs = run.gt_bufs[1:]
    finally:
        run.close()
    rec = {"workload": f"{n}^3 float volume x {total_views} views per dataset (view v on GPU v % N), {psf_edge}^3 PSF, inc={inc}, SNR {snr:g}, "
                       f"device-resident, one broadcast of the {4 * n ** 3 / 1e9:.1f} GB ground truth per step",
           "steps": steps, "ms_per_step": elapsed / steps * 1e3, "value": total_views * steps / elapsed * n ** 3 / 1e6, "unit": "Mvoxel/s",
           "views_per_s": total_views * steps / elapsed, "first_plane_mean_count": mean_count, "multi_gpu": diag}
    return rec, gt_keep


def tiled_leg(env, gt_dev, n, steps) -> dict:
    """BASELINE configs[3] as stated (n = 1024): 1024^3 volume, 6 views, anisotropic 31 x 31 x 63 PSF (sigma 2 / 2.2 / 12), 4x axial
    downsample, EVERY view cut into N z slabs -- one per rank -- through mvsim_view_slab_convolve_dev / _finish_dev
    (multiview-simulation_amd/tiling.py).  The halo planes are recomputed from the broadcast ground truth, the only exchange per view is
    the one double of adjustImage's sum, reduced by the C ABI's own mvsim_comm_allreduce_sum_f64 (torch.distributed only where RCCL
    cannot run: gloo ranks sharing a GPU)."""
    torch, dist, mvs, synth = env.torch, env.dist, env.mvs, env.synth
    tiling = importlib.import_module("multiview-simulation_amd.tiling")
    inc, views = 4, 6
    dims = (n, n, n)
    psfs = [synth.gaussian_psf(31, 31, 63, sigma=(2.0, 2.2, 12.0 + 0.05 * v)) for v in range(views)]
    reduce_fn, reduce_name = None, "ncclAllReduce of the device-resident double inside mvsim_view_slab_dev (C ABI, RCCL)"
    if env.bc_ctx is None:
        def reduce_fn(x):
            t = torch.tensor([x], dtype=torch.float64)
            if env.world > 1:
                dist.all_reduce(t)
            return float(t.item())
        reduce_name = f"torch.distributed.all_reduce ({env.backend}: rehearsal, RCCL cannot place two ranks on one device)"
    with mvs.Context(env.dev_index) as ctx:
        tv = tiling.TiledView(ctx, env.rank, env.world, comm_ctx=env.bc_ctx, allreduce_f64=reduce_fn)
        planes = tv.acq_planes(n, inc)
        acq = torch.empty(max(1, planes) * n * n, dtype=torch.float32, device=env.dev)
        params = [ctx.view_params(degrees=15 + 60 * v, inc=inc, snr=25.0, seed=464232194, stream=v, conv_method=1) for v in range(views)]
        info = []

        def step(record):
            for v in range(views):
                # the C ABI's communicator (or one rank): ONE asynchronous call per view and rank, the slab sum reduced on the device;
                # a reduction the harness brings (gloo ranks sharing a GPU): the three-step form, the sum through the host
                r = (tv.run_on_device if reduce_fn is None else tv.run)(gt_dev.data_ptr(), dims, psfs[v].copy(), params[v], acq.data_ptr())
                t0 = time.perf_counter()
                ctx.synchronize()                       # the next view reuses the slab workspace and `acq`
                r["finish_ms"] = (time.perf_counter() - t0) * 1e3
                if record:
                    info.append(r)

        def sync():
            torch.cuda.synchronize()
            if env.world > 1:
                dist.barrier()
            torch.cuda.synchronize()
        step(False)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        sync()
        own = time.perf_counter() - t0
        mean_count = float(acq[: n * n].double().mean().item()) if planes else None
    elapsed = own
    mine = torch.tensor([own / steps * 1e3, sum(r["convolve_ms"] for r in info) / len(info), sum(r["allreduce_ms"] for r in info) / len(info),
                         sum(r["finish_ms"] for r in info) / len(info), float(info[0]["planes_owned"]), float(info[0]["planes_rotated"])],
                        dtype=torch.float64, device=env.dev if env.backend == "nccl" else "cpu")
    every = [mine]
    if env.world > 1:
        tt = torch.tensor([own], dtype=torch.float64, device=mine.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        every = [torch.zeros_like(mine) for _ in range(env.world)]
        dist.all_gather(every, mine)
    rows = [[float(x) for x in t.tolist()] for t in every]
    return {"workload": f"BASELINE configs[3]: {n}^3 float volume, {views} views at 60-degree steps, 31x31x63 PSF, inc={inc} (4x axial downsample), "
                        f"SNR 25, every view tiled into {env.world} z slab(s), one per rank; device-resident, ground truth already broadcast",
            "steps": steps, "ms_per_step": elapsed / steps * 1e3, "ms_per_view": elapsed / (steps * views) * 1e3,
            "value": views * steps / elapsed * n ** 3 / 1e6, "unit": "Mvoxel/s", "first_plane_mean_count": mean_count,
            "reduction": reduce_name,
            "path": ("mvsim_view_slab_dev: one asynchronous call per view and rank, the slab sum reduced in place on the device (no host trip)"
                     if reduce_fn is None else "mvsim_view_slab_convolve_dev -> host reduction -> mvsim_view_slab_finish_dev"),
            "per_rank": [{"rank": i, "ms_per_step": round(r[0], 4), "slab_convolve_ms_per_view": round(r[1], 4), "allreduce_ms_per_view": round(r[2], 4),
                          "finish_ms_per_view": round(r[3], 4), "planes_owned": int(r[4]), "planes_rotated": int(r[5]),
                          "halo_recompute_share": round(1.0 - r[4] / r[5], 4) if r[5] > 0 else None} for i, r in enumerate(rows)],
            "note": "slab_convolve = rotate + attenuate of the slab and its halo planes, the convolution of the slab, the slab sum to the host; "
                    "allreduce = the one double of Tools.adjustImage's sum over the ranks (host clock around the call); finish = adjust, extract, "
                    "Poisson of the rank's acquired planes; halo_recompute_share = planes rotated beyond the rank's own / planes rotated"}


def dry_run_launch(args, world: int, rank: int) -> None:
    """Launcher / rendezvous check: no GPU, no libmvsim (tests/test_host_logic.py runs it with --backend gloo)."""
    import torch
    import torch.distributed as dist
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend if args.backend != "nccl" or torch.cuda.is_available() else "gloo",
                                rank=rank, world_size=world)
        t = torch.ones(1)
        if dist.get_backend() == "nccl":
            t = t.cuda(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run_launch": True, "n_gpus": world, "ranks_seen": seen, "backend": args.backend,
                          "self_launched": os.environ.get("MVSIM_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    if world > 1:
        dist.destroy_process_group()



BROADCAST_FORMS = ("scatter_allgather", "pipelined", "peer_copy")


def pick_broadcast(ab: dict, default: str) -> str:
    """Which form of the ground-truth broadcast the line reports: the one whose K timed steps took least (ms_per_step: MAX over ranks,
    the same number on every rank); a form that failed or has no number cannot win; ties and an empty table go to `default`."""
    best, best_ms = default, None
    if isinstance(ab.get(default), dict) and isinstance(ab[default].get("ms_per_step"), (int, float)) and "failed" not in ab[default]:
        best_ms = float(ab[default]["ms_per_step"])
    for form in BROADCAST_FORMS:
        rec = ab.get(form)
        if form == default or not isinstance(rec, dict) or "failed" in rec or not isinstance(rec.get("ms_per_step"), (int, float)):
            continue
        if best_ms is None or float(rec["ms_per_step"]) < best_ms:
            best, best_ms = form, float(rec["ms_per_step"])
    return best


def start_watchdog(out: dict, seconds: float, what: str):
    """Rank 0 of an N > 1 job: the line of record exists, what follows are collectives over every rank -- a rank lost inside one leaves
    the others waiting for ever.  If `what` has not come back after `seconds`, print the line as it stands and end the job (the launcher
    stops the other ranks).  A hung leg costs its leg, not the line."""
    import threading

    def give_up():
        out["legs_failed"] = f"{what} did not finish within {seconds:.0f} s (a rank lost inside a collective?)"
        print(ordered_line(out), flush=True)
        os._exit(4)
    t = threading.Timer(seconds, give_up)
    t.daemon = True
    t.start()
    return t


def ordered_line(out: dict) -> str:
    """The bench line with what a reader of the first 200 characters needs in front: value, unit, value_dense, the whole view's and the
    dominant stage's fraction of the HBM roofline on algorithmic bytes; everything else in its usual order."""
    roof = out.get("roofline") or {}
    lead = {"value": out.get("value"), "unit": out.get("unit"), "value_dense": out.get("value_dense"),
            "whole_view_frac": (roof.get("whole_view") or {}).get("frac"), "roofline_frac": roof.get("frac")}
    lead = {k: (round(v, 4) if isinstance(v, float) and k.endswith("frac") else v) for k, v in lead.items()}
    return json.dumps({**lead, **{k: v for k, v in out.items() if k not in lead}})

def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  Nothing in this process has imported torch or touched a GPU yet.
        raise SystemExit(self_launch(args))
    if os.environ.get("MVSIM_BENCH_SELF_LAUNCHED") == "1":
        die_with_parent()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (self-launching) "
                         f"or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    backend_asked = args.backend
    if args.backend == "auto":
        args.backend = "gloo"
    if args.dry_run_launch:
        return dry_run_launch(args, world, rank)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    collective = args.collective
    if collective == "auto":
        # the C ABI's communicator wherever every rank has a device of its own; an explicit `--backend gloo` with ranks sharing a GPU is
        # a rehearsal of the control flow (RCCL cannot place two ranks on one device): torch.distributed moves the bytes there
        collective = "mvsim" if (world <= torch.cuda.device_count() and (backend_asked != "gloo" or world == 1)) else "torch"
    multi = world > 1 or args.rehearse_multi              # the N > 1 data path (broadcast per step, double-buffered ground truth)
    if args.rehearse_multi and world == 1:
        collective = "mvsim"

    mvs = importlib.import_module("multiview-simulation_amd")
    synth = importlib.import_module("multiview-simulation_amd.synthetic")
    build = importlib.import_module("multiview-simulation_amd.build")

    env = Env()
    env.torch, env.dist, env.mvs, env.synth = torch, dist, mvs, synth
    env.dev, env.dev_index, env.world, env.rank = dev, dev_index, world, rank
    env.multi, env.collective, env.backend, env.broadcast = multi, collective, args.backend, args.broadcast
    env.bc_ctx, env.bc_stream = None, None

    n = args.size
    dims = (n, n, n)
    nvox = n ** 3
    total_views = args.views_total if args.scaling == "strong" else args.views_per_gpu * world
    nzo = (n - 1) // args.inc + 1

    # every rank really is there: one all-reduce of a 1 through torch.distributed, and (below) one through the C ABI's own
    # communicator
    ranks_seen = {"torch": 1}
    if world > 1:
        t = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t)
        ranks_seen["torch"] = int(t.item())
        if ranks_seen["torch"] != world:
            raise SystemExit(f"rank {rank}: all-reduce of 1 over the process group gave {ranks_seen['torch']}, expected {world}")
    if multi:
        env.bc_stream = torch.cuda.Stream(device=dev)
        if collective == "mvsim":
            # the C ABI's own RCCL communicator: rank 0 creates the id, torch.distributed is only the messenger
            box = [mvs.Context.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            env.bc_ctx = mvs.Context(dev_index)
            env.bc_ctx.set_stream(env.bc_stream.cuda_stream)
            env.bc_ctx.set_option("broadcast", args.broadcast)
            env.bc_ctx.comm_init(world, rank, box[0])
            one = torch.ones(16, dtype=torch.float32, device=dev)
            with torch.cuda.stream(env.bc_stream):
                env.bc_ctx.comm_allreduce_sum(one.data_ptr(), 16)
            env.bc_stream.synchronize()
            ranks_seen["mvsim_comm"] = int(one[0].item())
            if ranks_seen["mvsim_comm"] != world:
                raise SystemExit(f"rank {rank}: all-reduce of 1 over the C ABI's communicator gave {ranks_seen['mvsim_comm']}, expected {world}")
    bc_ctx = env.bc_ctx

    # synthetic inputs (rank 0 owns the ground truth; other ranks receive it by broadcast, once per step = dataset)
    gt_host = synth.sphere_phantom(n) if rank == 0 else None
    psf_raw = synth.gaussian_psf(args.psf, sigma=(2.0, 2.2, 6.0))

    def fill_gt(buf):
        buf.copy_(torch.from_numpy(gt_host.reshape(-1)))
    run = ShardedRun(env, n, (args.psf,) * 3, (2.0, 2.2, 6.0), args.inc, args.snr, args.conv_method, total_views, args.streams,
                     serial=args.serial, serial_broadcast=args.serial_broadcast, fill_gt=fill_gt)
    my_views, angles, ctxs, gt_bufs, acq, psfs, params = run.my_views, run.angles, run.ctxs, run.gt_bufs, run.acq, run.psfs, run.params
    ctx = ctxs[0]
    step, sync, set_overlap = run.step, run.sync, run.set_overlap

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1 and args.warmup > 0:
        run.check_broadcast()                     # outside the timed region: every rank holds rank 0's ground truth
    # ---- the timed region: exactly K steps between two barrier + synchronise pairs; MAX over ranks below
    timed_with_events = args.serial or multi      # serial main line: its stage events ARE the roofline's source
    if timed_with_events:
        for c in ctxs:
            c.enable_timing(True)
    elapsed, multi_diag = run.timed(args.steps)

    def read_stage():
        acc = {}
        for c in ctxs:
            for k, v in c.timings().items():
                acc[k] = acc.get(k, 0.0) + v / len(ctxs)
        return acc
    stage = None
    stage_overlapped = not args.serial
    if timed_with_events:
        if rank == 0 and my_views:
            stage = read_stage()
        for c in ctxs:
            c.enable_timing(False)

    serial_leg = None
    if rank == 0 and not multi and not args.serial and my_views:
        # the roofline's source: the same K steps with both overlaps off -- one kernel at a time on the context's stream, HIP
        # events around every stage (on that stream), so that stage times are the kernels' own durations
        set_overlap(False)
        step(); sync()
        for c in ctxs:
            c.enable_timing(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t1
        stage = read_stage()
        stage_overlapped = False
        for c in ctxs:
            c.enable_timing(False)
        serial_leg = {"ms_per_step": dt / args.steps * 1e3, "value": total_views * args.steps / dt * nvox / 1e6, "unit": "Mvoxel/s",
                      "note": "options tail_overlap = psf_overlap = 0 (kernels strictly one at a time): the leg `roofline` is read from"}
        set_overlap(True)

    plane_stats = ctx.plane_stats() if (rank == 0 and my_views) else None      # of the leg `roofline` is read from (before the dense leg)
    queue_stats = ctx.queue_stats() if (rank == 0 and my_views and args.snr >= 0) else None     # the sampler's work queue after the timed steps
    no_empty = None
    if rank == 0 and not multi and my_views and args.conv_method == 1 and not args.no_dense_leg:
        # The rotate + attenuate + x-transform kernel skips the fp64 blends and the transforms of rows that hold no non-zero voxel
        # (exact; the phantom, like the reference's drawSpheres volume, is empty outside the specimen).  How much of `value` is
        # owed to that: the same K serial steps on the SAME phantom plus 1e-6 everywhere -- no empty row left, the same intensity
        # distribution (adjustImage adds 1e-4 to every voxel anyway), so the sampler sees the same regime mix.
        keep = gt_bufs[0].clone()
        gt_bufs[0].add_(1e-6)
        set_overlap(False)
        step(); sync()
        for c in ctxs:
            c.enable_timing(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t1
        st = read_stage()
        for c in ctxs:
            c.enable_timing(False)
        no_empty = {"ms_per_step": dt / args.steps * 1e3, "value": total_views * args.steps / dt * nvox / 1e6, "unit": "Mvoxel/s",
                    "rotate_attenuate_ms": round(st["rotate_ms"] + st["attenuate_ms"], 4), "extract_ms": round(st["extract_ms"], 4),
                    "note": "serial leg on the phantom + 1e-6 in every voxel: no empty rows for the zero fast paths of "
                            "k_rotate_attenuate_fftx (DESIGN.md 4.1) to skip; same intensity distribution"}
        gt_bufs[0].copy_(keep)
        del keep
        set_overlap(not args.serial)

    compact_queue = None
    if rank == 0 and not multi and my_views and args.snr >= 0 and queue_stats and not args.no_compact_queue_leg:
        # The same K steps (library defaults) with the sampler's work queue on its automatic share (option poisson_queue_share=auto): segments
        # of 5 sixteenths of their blocks' voxels at first, more once a view has needed more; the appends check for room, a third kernel
        # looks for refused voxels.  What the smaller queue costs in time, next to what it saves in memory.
        for c in ctxs:
            c.synchronize()
            c.set_option("poisson_queue_share", "auto")
            c.release_caches()                                  # the queue is only ever grown: start it again
        step(); step(); sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = time.perf_counter() - t1
        q = ctx.queue_stats()
        compact_queue = {"value": total_views * args.steps / dt * nvox / 1e6, "unit": "Mvoxel/s", "ms_per_step": dt / args.steps * 1e3,
                         "gib": round(q["bytes"] / 2 ** 30, 3), "segment_items": q["segment_items"], "refused_voxels": q["refused"],
                         "note": "poisson_queue_share=auto; counts identical to the default's (tests/test_gpu_parity.py)"}
        for c in ctxs:
            c.set_option("poisson_queue_share", 16)

    if rank == 0 and args.rehearse_multi:
        # the rehearsal's own check: both ground-truth buffers still hold the phantom, the views produced counts
        assert all(torch.equal(b, gt_bufs[0]) for b in gt_bufs) and float(gt_bufs[0].max()) > 0 and float(acq[-1].max()) > 0
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        views_s = total_views * args.steps / elapsed
        mvox_s = views_s * nvox / 1e6
        conv_name = ("fft (hand-written LDS FFT passes in x and y, direct Kz-tap convolution in z; rocFFT only for unsupported sizes)"
                     if args.conv_method == 1 else "direct LDS-tiled stencil (fp32-FMA-bound)" if args.conv_method == 2 else "auto")
        out = {
            "metric": "simulated Mvoxel/s (views x input voxels / s), 512^3 volume x 8 views",
            "value": mvox_s, "unit": "Mvoxel/s", "views_per_s": views_s,
            # `value` is measured on the sphere phantom -- a specimen in empty space, like the reference's own drawSpheres volume
            # (SimulateMultiViewDataset.java:436-522) -- whose empty rows and planes the kernels skip exactly; `value_dense` is the
            # same workload without a single empty voxel (phantom + 1e-6; serial leg, see `no_empty_space`).  Read them side by side.
            "value_dense": (no_empty["value"] if no_empty else None),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            **({"rehearsal": "N > 1 data path on one GPU (--rehearse-multi): a control-flow check, not a result"} if args.rehearse_multi else {}),
            "dtype": "f32 (f64 attenuation/reductions/Poisson)", "data": "synthetic",
            "config": {"workload": f"{n}^3 float volume x {total_views} views per dataset (one dataset per step), {args.psf}^3 PSF, "
                                   f"rotate+attenuate+convolve+adjust+extract(inc={args.inc})+Poisson(SNR {args.snr:g}), "
                                   f"device-resident; BASELINE configs[1] per view, configs[2] sharding (view v on GPU v % N)",
                       "volume": [n, n, n], "psf": [args.psf] * 3, "views_total": total_views,
                       "views_this_gpu": len(my_views), "inc": args.inc, "snr": args.snr,
                       "conv_method": conv_name,
                       "streams_per_gpu": len(ctxs),
                       "overlap": ("off (--serial)" if args.serial else
                                   "library defaults: psf_overlap (PSF spectrum on a side stream beside the image's first passes) + tail_overlap "
                                   "(extract + Poisson of view v beside the first kernel of view v+1 -- used only where the separate rotate kernel "
                                   "runs: beside the fused rotate + attenuate + x-transform kernel it measured slower and is skipped); "
                                   "bit-identical to the serial order"),
                       "launcher": ("self-launched by bench.py (one child process per rank)" if os.environ.get("MVSIM_BENCH_SELF_LAUNCHED") == "1"
                                    else "external launcher (torchrun)" if world > 1 else "single process"),
                       "ranks_seen": ranks_seen,
                       "rccl": rccl_report(mvs, torch),
                       "collective": ("none" if not multi else
                                      (f"mvsim_comm_broadcast_volume ({args.broadcast}, RCCL over xGMI)" if bc_ctx is not None
                                       else f"torch.distributed.broadcast ({args.backend})")
                                      + ", one per step, " + ("serial" if args.serial_broadcast else "issued one dataset ahead"))},
        }
        if multi_diag:
            out["multi_gpu"] = multi_diag
        if serial_leg:
            out["serial"] = serial_leg
        if no_empty:
            out["no_empty_space"] = no_empty
        if queue_stats:
            # the Poisson work queue of this context: per-block segments that hold every voxel of their blocks (16 B per acquired
            # voxel), what the last view queued, the fullest block; `auto_share`: the same steps with segments sized from what the views need
            q = queue_stats
            out["poisson_queue"] = {"share": "16/16 (default: every voxel of a block fits its segment)", "gib": round(q["bytes"] / 2 ** 30, 3), "segment_items": q["segment_items"], "queued_share_of_voxels": round((q["bright"] + q["inversion"]) / (n * n * nzo), 4),
                                    "fullest_block_pending": q["fullest_block"], "refused_voxels": q["refused"]}
            if compact_queue:
                out["poisson_queue"]["auto_share"] = compact_queue
        if stage:
            kernel_sha = build.source_sha()
            traffic, note = load_traffic(n, args.psf, args.inc, len(ctxs), args.conv_method, kernel_sha)
            wall_ms = (serial_leg["ms_per_step"] if serial_leg else ms_per_step)
            wall_view = wall_ms / max(1, len(my_views)) if len(ctxs) == 1 else None
            out["roofline"] = roofline_record(mvs, stage, nvox, n * n * nzo, n, args.psf, args.conv_method, traffic, note,
                                              view_wall_ms=wall_view, overlapped=stage_overlapped, plane_stats=plane_stats)
            out["kernel_sha"] = kernel_sha
    if rank == 0 and not multi and len(ctxs) == 1 and not args.no_main_iteration and args.conv_method == 1 and my_views:
        # The whole body of `main`'s view loop (SimulateMultiViewDataset.java:567-613): the view, then makeIsotropic and the
        # rotate-backs of the isotropic view, the weight image and the PSF (mvsim_simulate_iteration_dev), all outputs in HBM.
        try:
            niso = (nzo - 1) * args.inc + 1
            extra = {"iso": torch.empty(n * n * niso, dtype=torch.float32, device=dev), "view": torch.empty(n * n * niso, dtype=torch.float32, device=dev),
                     "w": torch.empty(nvox, dtype=torch.float32, device=dev), "psf": torch.empty(args.psf ** 3, dtype=torch.float32, device=dev)}

            def iterations():
                for i in range(len(my_views)):
                    ctx.simulate_iteration_dev(gt_bufs[0].data_ptr(), dims, psfs[i].copy(), params[i], -angles[my_views[i]], acq[i].data_ptr(),
                                               iso_dptr=extra["iso"].data_ptr(), view_dptr=extra["view"].data_ptr(),
                                               view_weights_dptr=extra["w"].data_ptr(), view_psf_dptr=extra["psf"].data_ptr())
            iterations(); sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                iterations()
            sync()
            dt = time.perf_counter() - t1
            out["main_iteration"] = {"ms_per_iteration": dt / (args.steps * len(my_views)) * 1e3,
                                     "value": total_views * args.steps / dt * nvox / 1e6, "unit": "Mvoxel/s",
                                     "note": "one iteration = the view (rotate .. Poisson) + makeIsotropic + rotateAroundAxis(iso, -angle) + "
                                             "rotateAroundAxis(computeWeightImage, -angle) + rotateAroundAxis(psf, -angle), device-resident "
                                             "(SimulateMultiViewDataset.java:567-613 without the TIFF writes); same workload as `value`"}
            del extra
        except Exception as e:
            out["main_iteration"] = {"failed": repr(e)}
    if rank == 0 and not multi and len(ctxs) == 1 and not args.no_two_streams and args.conv_method == 1:
        # two contexts on the GPU: views alternate between them (reported beside `value`)
        def timed_steps():
            for _ in range(max(1, args.warmup)):
                step()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            dt = time.perf_counter() - t0
            return {"ms_per_step": dt / args.steps * 1e3, "value": total_views * args.steps / dt * nvox / 1e6, "unit": "Mvoxel/s"}
        try:
            ctxs.append(mvs.Context(dev_index))
            set_overlap(not args.serial)
            out["two_streams"] = dict(timed_steps(), streams_per_gpu=2,
                                      note="views alternate between two contexts of the same GPU; same workload and timed-region rules as `value`")
        except Exception as e:
            out["two_streams"] = {"failed": repr(e)}
    legs = {}
    legs_error = None
    if multi and bc_ctx is not None and not args.no_broadcast_ab and my_views is not None:
        # The forms of the ground-truth broadcast, back to back in THIS invocation: the line's own K steps used
        # --broadcast's form; now every other form runs the same K steps between the same barrier + synchronise pairs.  All of them move
        # the same bytes into the same buffers (check_broadcast after each), so which one the line reports is a tuning choice, disclosed in
        # multi_gpu.broadcast_ab: the fastest.  Each form under a watchdog of its own: a form that hangs costs its leg -- rank 0 prints the
        # line with what it has -- not the line.
        def ab_entry(el, dg):
            return {"ms_per_step": round(el / args.steps * 1e3, 4), "value": total_views * args.steps / el * nvox / 1e6,
                    "broadcast_ms": (dg or {}).get("broadcast_ms"), "views_ms": (dg or {}).get("views_ms"),
                    "broadcast_hidden_frac": (dg or {}).get("broadcast_hidden_frac")}
        ab = {args.broadcast: ab_entry(elapsed, multi_diag)}
        diags = {args.broadcast: multi_diag}
        times = {args.broadcast: elapsed}
        if rank == 0:
            out.setdefault("multi_gpu", {})["broadcast_ab"] = ab
        for form in BROADCAST_FORMS:
            if form == args.broadcast or legs_error:
                continue
            wd = start_watchdog(out, 90.0, f"the broadcast form '{form}' (multi_gpu.broadcast_ab)") if (world > 1 and rank == 0) else None
            try:
                run.set_broadcast(form)
                run.step(); run.step(); run.sync()            # both ground-truth buffers through this form before anything is timed
                if world > 1:
                    run.check_broadcast()
                el2, dg2 = run.timed(args.steps)
                ab[form], diags[form], times[form] = ab_entry(el2, dg2), dg2, el2
            except BaseException as e:                         # (check_broadcast raises SystemExit)
                ab[form] = {"failed": repr(e)}
                if world > 1:
                    legs_error = f"rank {rank}: broadcast form {form}: {e!r}"     # the ranks may be out of step now: nothing collective after this
            if wd is not None:
                wd.cancel()
        chosen = pick_broadcast(ab, args.broadcast)
        if not legs_error:
            run.set_broadcast(chosen)                          # the legs behind the line use the chosen form as well
        if rank == 0:
            el = times[chosen]
            out["value"] = total_views * args.steps / el * nvox / 1e6
            out["views_per_s"] = total_views * args.steps / el
            out["ms_per_step"] = el / args.steps * 1e3
            out["multi_gpu"] = dict(diags[chosen] or {}, broadcast_ab=ab, broadcast_chosen=chosen,
                                    broadcast_note="every form ran the same K steps in this invocation (same buffers, same views, checked after "
                                                   "each); value / ms_per_step / the diagnostics above are the fastest form's")
            out["config"]["collective"] = out["config"]["collective"].replace(f"({args.broadcast},", f"({chosen}: fastest of the forms in multi_gpu.broadcast_ab,")
    gt512 = gt_bufs[0]                                     # rank 0's phantom (every rank's after a broadcast): the 1024^3 legs up-sample it
    run.close()
    watchdog = None
    if world > 1 and rank == 0:
        # The legs below are collectives over every rank: a rank that dies inside one leaves the others waiting.  The line of record
        # exists already; if the legs have not come back after five minutes, rank 0 prints it without them and ends the job.
        watchdog = start_watchdog(out, 300.0, "the legs behind the line of record")
    if multi and args.conv_method == 1 and n <= 512 and not legs_error:
        # the other sizes north_star names, on the N > 1 data path: 1024^3 views sharded v % N (`size_1024`) and BASELINE configs[3] as
        # stated -- every 1024^3 view cut into N z slabs (`tiled_1024`).  Every rank takes part; rank 0 reports.  (The legs run at twice
        # the main line's edge: 1024 for the 512^3 line of record, 512 for the tests' 256^3 rehearsals.)
        leg_steps = max(1, min(args.steps, 3))
        n2 = 2 * n
        gt2 = None
        try:
            if not args.no_size_1024:
                legs[f"size_{n2}"], gt2 = sharded_2x_leg(env, gt512, n, args.psf, args.inc, args.snr, total_views, leg_steps, args.serial_broadcast)
            if not args.no_tiled_1024:
                if gt2 is None:
                    gt2 = torch.empty(n2 ** 3, dtype=torch.float32, device=dev)
                    if rank == 0:
                        gt2.copy_(upsample2x(torch, gt512, n))
                    if world > 1:
                        if bc_ctx is not None:
                            if env.broadcast == "peer_copy":
                                bc_ctx.comm_register_volume(gt2.data_ptr(), n2 ** 3)
                            with torch.cuda.stream(env.bc_stream):
                                bc_ctx.comm_broadcast_volume(gt2.data_ptr(), n2 ** 3, 0)
                            env.bc_stream.synchronize()
                            if env.broadcast == "peer_copy":
                                bc_ctx.comm_unregister_volume(gt2.data_ptr())
                        else:
                            dist.broadcast(gt2, src=0)
                    torch.cuda.synchronize()
                legs[f"tiled_{n2}"] = tiled_leg(env, gt2, n2, leg_steps)
        except Exception as e:
            # The line of record is complete at this point and must not be lost to a leg: rank 0 still prints it (below), with the
            # failure in it; with N > 1 the job then ends with a non-zero status (a rank that dropped out of a collective must end
            # the job, not leave the others waiting).
            legs_error = f"rank {rank}: {e!r}"
            legs.setdefault(f"size_{n2}" if f"size_{n2}" not in legs and not args.no_size_1024 else f"tiled_{n2}", {"failed": repr(e)})
        del gt2
    if watchdog is not None:
        watchdog.cancel()
    del gt512
    torch.cuda.empty_cache()
    if bc_ctx is not None:
        bc_ctx.close()
    if rank == 0:
        out.update(legs)
    if rank == 0:
        if not multi and not args.no_end_to_end and args.conv_method == 1:
            try:
                out["end_to_end"] = end_to_end_record(mvs, dev_index, gt_host, [psf_raw], angles, args.inc, args.snr)
            except Exception as e:  # reported extras never cost the GPU line
                out["end_to_end"] = {"failed": repr(e)}
        if not multi and not args.no_size_1024 and n == 512 and args.conv_method == 1:
            try:
                g512 = torch.from_numpy(gt_host.reshape(-1)).to(dev)
                out["size_1024"] = size_1024_record(mvs, torch, dev, dev_index, g512, psf_raw, args.inc, args.snr)
                del g512
            except Exception as e:
                out["size_1024"] = {"failed": repr(e)}
        if not multi and not args.no_small_views and args.conv_method == 1:
            try:
                out["small_views"] = small_views_record(mvs, synth, dev_index)
            except Exception as e:
                out["small_views"] = {"failed": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(gt_host, psf_raw, angles[0], args.inc, args.snr, args.cpu_slab, args.cpu_poisson_planes)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "Mvoxel/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        if legs_error:
            out["legs_failed"] = legs_error
        print(ordered_line(out), flush=True)

    if legs_error and world > 1:
        sys.stderr.write(f"bench.py: a leg behind the line of record failed ({legs_error}); ending the job\n")
        sys.stderr.flush()
        os._exit(3)                                        # the other ranks may be inside a collective: no orderly teardown
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: image-pairs/sec of SuperPoint(view0) + SuperPoint(view1) + LightGlue forward.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload = BASELINE.json config 2: 640x480 gray images, 1024 keypoints, 32 pairs per GPU per step
(weak scaling: every rank runs its own 32 pairs; the only collective is one RCCL all-gather of the
match outputs per step).  Inputs are synthetic and resident in HBM before the timed region: SURVEY 8(d)'s
tiling of the boat pair (the reference's test asset, carried by tests/golden/two_view_boat.npz) resized to
the workload size with per-item circular shifts; `--images noise` gives seeded smooth noise instead.
Weights are seeded (there are no checkpoints offline).  One "step" = one pass of the hot path over one
batch: extractor on 2B images, matcher on B pairs, all-gather.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, one child
process per device, before the parent makes any HIP call (the pattern of gluefactory/train.py:699-701);
under `torch.distributed.run` the ranks are already there.  It never reports fewer GPUs than asked for.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel, HIP-event
timed inside the timed region), `f32_mode` (the same workload on the fp32-input MFMA path, timed in the
same run) and, at N == 1, `cpu_baseline` (the CPU oracle, SURVEY 8(d) protocol, on a bounded sample).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "glue-factory_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402  (importing torch does not initialise HIP; the launcher parent makes no torch.cuda call at all)
import torch.distributed as dist  # noqa: E402

H, W, K, PAIRS = 480, 640, 1024, 32
# TFLOP/s, MI355X_MICROARCH.md: dense MFMA peak of the dtype the matrix pipe is fed with
# sustained rate of the same MFMA with operands of realistic bit activity (tests/micro/mfma_peak.hip, profiles/*_micro.txt):
# the chip power-throttles from 2.37 to ~1.6 GHz, so 2.5 PF is only reached with near-constant operands
SUSTAINED = {"f32": None, "f16x3": 1668.0}
PEAK = {"f32": 157.3,      # v_mfma_f32_32x32x2_f32
        "f16x3": 2500.0}   # v_mfma_f32_32x32x16_f16; every algorithmic product costs 3 MFMA products (hi.hi + hi.lo + lo.hi)


def boat_images(n, seed):
    """boat1 / boat2 resized bilinearly to the workload size, tiled over the batch with per-item integer
    circular shifts (natural image statistics).  Even items come from boat1, odd items from boat2."""
    import numpy as np

    z = np.load(os.path.join(ROOT, "tests", "golden", "two_view_boat.npz"))
    g = torch.Generator().manual_seed(seed)
    base = [torch.nn.functional.interpolate(torch.from_numpy(z[k]).float()[None, None] / 255.0, size=(H, W), mode="bilinear", align_corners=False)[0, 0]
            for k in ("gray0", "gray1")]
    out = torch.empty(n, 1, H, W)
    for i in range(n):
        dy, dx = (int(v) for v in torch.randint(0, 10 ** 6, (2,), generator=g))
        out[i, 0] = torch.roll(base[i & 1], shifts=(dy % H, dx % W), dims=(0, 1))
    return out.contiguous()


def make_images(kind, n_pairs, seed):
    """(2*n_pairs, 1, H, W): the view-0 images of all pairs, then the view-1 images."""
    if kind == "noise":
        return synthetic_images(2 * n_pairs, seed)
    im = boat_images(2 * n_pairs, seed)  # item 2i = boat1 shifted, 2i+1 = boat2 shifted -> pair i = (2i, 2i+1)
    return torch.cat([im[0::2], im[1::2]], 0).contiguous()


def synthetic_images(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 1, H + 8, W + 8, generator=g)
    x = torch.nn.functional.avg_pool2d(x, 5, 1)[:, :, 2:H + 2, 2:W + 2]
    x = x - x.amin(dim=(2, 3), keepdim=True)
    return (x / x.amax(dim=(2, 3), keepdim=True)).contiguous()


def algorithmic_flops(B):
    """Algorithmic FLOPs per launch of every profiled kernel family (SURVEY.md section 8d formulae);
    B = pairs per step; SuperPoint runs on 2B images."""
    n = 2 * B
    hw = [H * W, H * W // 4, H * W // 16, H * W // 64]
    conv = lambda ci, co, px: 2.0 * 9 * ci * co * px * n
    T = (K + K) * B  # tokens per step
    lin = lambda i, o: 2.0 * T * i * o
    f = {
        "sp.conv1a": conv(1, 64, hw[0]), "sp.conv1b": conv(64, 64, hw[0]),
        "sp.conv1ab": conv(1, 64, hw[0]) + conv(64, 64, hw[0]),  # conv1a recomputed inside conv1b's tiles (f16x3 mode)
        "sp.conv2a": conv(64, 64, hw[1]), "sp.conv2b": conv(64, 64, hw[1]),
        "sp.conv3a": conv(64, 128, hw[2]), "sp.conv3b": conv(128, 128, hw[2]),
        "sp.conv4a": conv(128, 128, hw[3]), "sp.conv4b": conv(128, 128, hw[3]),
        "sp.convPa": conv(128, 256, hw[3]), "sp.convDa": conv(128, 256, hw[3]),
        "sp.convPb": 2.0 * 256 * 65 * hw[3] * n, "sp.convDb": 2.0 * 256 * 256 * hw[3] * n,
        "lg.gemm.qkv": lin(256, 768), "lg.gemm.qkv_cross": lin(256, 512), "lg.gemm.out": lin(256, 256),
        "lg.gemm.ffn0": lin(512, 512), "lg.gemm.ffn3": lin(512, 256), "lg.gemm.final_proj": lin(256, 256),
        "lg.ffn": lin(512, 512) + lin(512, 256),  # ffn.0 + ffn.3 in one launch (csrc/ffn_f16.hip)
        "sp.heads": 2.0 * 256 * 65 * hw[3] * n,    # convPb inside the fused detector head (csrc/heads_f16.hip)
        "sg.ffn": lin(512, 512) + lin(512, 256), "gs.ffn": lin(512, 512) + lin(512, 256),
        "lg.gemm.sim": 2.0 * B * K * K * 256,
        # attention contractions: 2*h*Nq*Nk*dh each; self = QK^T + PV on both images;
        # cross = one shared similarity + two PV products (the reference computes sim once, lightglue.py:201)
        "lg.attn.self": 2 * (2.0 * 4 * K * K * 64) * 2 * B,
        "lg.attn.cross": 3 * (2.0 * 4 * K * K * 64) * B,
        # SuperGlue (BASELINE.json configs[3]): same linear shapes; cross layers attend in both directions
        # with separate similarities (superglue.py:118-141), so self and cross cost the same
        "sg.gemm.qkv": lin(256, 768), "sg.gemm.mlp0": lin(512, 512), "sg.gemm.mlp3": lin(512, 256),
        "sg.gemm.final_proj": lin(256, 256), "sg.gemm.cost": 2.0 * B * K * K * 256,
        "sg.attn.self": 2 * (2.0 * 4 * K * K * 64) * 2 * B, "sg.attn.cross": 2 * (2.0 * 4 * K * K * 64) * 2 * B,
        "gs.gemm.qkv": lin(256, 768), "gs.gemm.mlp0": lin(512, 512), "gs.gemm.mlp3": lin(512, 256),
        "gs.attn.self": 2 * (2.0 * 4 * K * K * 64) * 2 * B, "gs.attn.cross": 2 * (2.0 * 4 * K * K * 64) * 2 * B,
    }
    return f


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """(threads to use, cores visible, where the number comes from): the affinity mask, cut to the cgroup CPU quota when one is
    set (the GPU boxes of this pool: `cpu.max` = "1600000 100000" = 16 CPUs of a 256-thread host -- more threads than that only
    time-slice each other: one batch-8 pass took 177 s on 256 threads against 4.8 s on 16); without a quota the count is capped at
    MG_CPU_CORES (default 16) and cpu_baseline() times an all-visible-cores leg as well."""
    seen = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = seen
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota not in ("max", "-1"):
                return max(1, min(n, int(int(quota) / int(period)))), seen, f"cgroup CPU quota ({path}: {quota} / {period})"
        except (OSError, ValueError, IndexError):
            continue
    return min(n, int(os.environ.get("MG_CPU_CORES", "16"))), seen, "no cgroup quota: default share (MG_CPU_CORES)"


def cores_evidence():
    """What the host says about this process's CPU share (read, not interpreted): cgroup quota, cpuset, pressure, load."""
    ev = {}
    for key, path in (("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max"), ("cpuset_effective", "/sys/fs/cgroup/cpuset.cpus.effective"),
                      ("pressure_cpu", "/proc/pressure/cpu"), ("loadavg", "/proc/loadavg")):
        try:
            ev[key] = open(path).read().strip().splitlines()[0][:120]
        except (OSError, IndexError):
            ev[key] = None
    return ev


def all_cores_leg(matcher, images, limit=45.0):
    """The leg of the CPU baseline on ALL visible cores, needed only where no cgroup quota says what this process may use.  It runs in a
    child process with a hard time limit (on a box whose share of the host is smaller than what it sees, hundreds of threads time-slice
    each other: 177 s for one batch-8 pass on 256 threads against 4.8 s on 16) -- and main() calls this BEFORE its first torch.cuda call:
    a process that has initialised HIP starts no child on this pool.  Returns a leg dict, or None when a quota defines the share.  No
    failure of the child (refused spawn, time-out, garbage) may cost the run its JSON line: it is recorded in the leg instead."""
    import subprocess

    cores, cores_seen, cores_source = usable_cores()
    if cores_seen <= cores or cores_source.startswith("cgroup"):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", f"{cores_seen},1", "--matcher", matcher, "--images", images,
           "--kpts", str(K), "--size", f"{W}x{H}"]
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=limit).stdout.strip().splitlines()
        return json.loads(out[-1])
    except Exception as e:  # TimeoutExpired, OSError (spawn refused), IndexError / ValueError (no JSON line)
        why = (f"did not complete one warm-up pass + one timed pass within {limit:.0f} s ({time.perf_counter() - t0:.0f} s waited): "
               f"slower than {1.0 / (limit / 2):.3f} pairs/s, i.e. below the {cores}-thread legs") if isinstance(e, subprocess.TimeoutExpired) \
            else f"leg not run: {type(e).__name__}: {e}"
        return {"batch": 1, "threads": cores_seen, "pairs_per_s": 0.0, "median_s": None, "reps": 0, "warmup": 0, "note": why}


def cpu_baseline(matcher="lightglue", images="boat", budget_s=60.0, note=lambda m: None, only=None, extra_leg=None):
    """The CPU oracle (oracle/sp_lg_oracle.py, a restatement of the reference) on a bounded sample of the same workload,
    SURVEY 8(d) protocol: B = 1 and B = 8, 3 warm-up + >= 10 timed repetitions each, time.perf_counter, median, on the threads
    `usable_cores()` gives: the cgroup CPU quota where one is set (it IS the box's share; stated in `cores_share_source`), else a 16-core
    share plus a leg on ALL visible cores (`extra_leg`, timed by `all_cores_leg` in a time-limited child process before the GPU work), so
    that the reported value is the best the host does either way (`value` = the fastest leg, `cores` = its thread count).  A repetition count is cut
    (and said so in `sample`) only if the leg would pass its budget."""
    from oracle import sp_lg_oracle as O

    torch.set_grad_enabled(False)
    cores, cores_seen, cores_source = usable_cores()
    W_sp = O.make_sp_weights(0)
    if matcher == "gluestick":
        return None  # secondary workload: its line inputs are built from the GPU run's keypoints
    W_lg = O.make_lg_weights(7, final_gain=8.0, match_bias=2.0) if matcher == "lightglue" else O.make_sg_weights(9)
    spc = dict(nms_radius=3, detection_threshold=0.0, max_num_keypoints=K)

    def run(batch, cores, budget_s=budget_s):
        torch.set_num_threads(cores)
        im = make_images(images, batch, 123)
        size = torch.tensor([[float(W), float(H)]] * batch)
        data = {"view0": {"image": im[:batch], "image_size": size}, "view1": {"image": im[batch:], "image_size": size}}

        def once():
            if matcher == "lightglue":
                O.two_view_forward(data, W_sp, W_lg, spc, dict(filter_threshold=0.1))
            else:
                p0, p1 = O.superpoint_forward(data["view0"], W_sp, **spc), O.superpoint_forward(data["view1"], W_sp, **spc)
                O.superglue_forward({**{k + "0": v for k, v in p0.items()}, **{k + "1": v for k, v in p1.items()}, **data}, W_lg)

        t_leg = time.perf_counter()
        warm = 0
        for _ in range(3):
            once()
            warm += 1
            if time.perf_counter() - t_leg > budget_s / 4:
                break
        per_rep = (time.perf_counter() - t_leg) / warm
        note(f"cpu oracle, batch {batch}, {cores} threads: {warm} warm-up run(s), {per_rep:.2f} s each")
        times = []
        while len(times) < 10:
            t0 = time.perf_counter()
            once()
            times.append(time.perf_counter() - t0)
            if len(times) % 3 == 0:
                note(f"cpu oracle, batch {batch}: {len(times)} timed repetitions")
            if time.perf_counter() - t_leg + per_rep > budget_s and len(times) >= (3 if per_rep < budget_s / 8 else 1):
                break
        med = statistics.median(times)
        return {"batch": batch, "threads": cores, "pairs_per_s": round(batch / med, 4), "median_s": round(med, 4), "reps": len(times), "warmup": warm}

    if only is not None:  # child process of the all-cores leg (all_cores_leg)
        return run(only[1], only[0], budget_s=budget_s)
    legs = [run(1, cores), run(8, cores)]
    if extra_leg is not None:  # the all-visible-cores leg, timed by main() BEFORE this process initialised HIP
        legs.append(extra_leg)
        note(f"cpu oracle, all {cores_seen} visible cores (timed before the GPU run): {extra_leg}")
    best = max(legs, key=lambda l: l["pairs_per_s"])
    return {"value": best["pairs_per_s"], "unit": "pairs/s", "cores": best["threads"], "cores_visible": cores_seen, "cores_share": cores,
            "cores_share_source": cores_source, "cores_evidence": cores_evidence(), "kind": "port", "cpu": cpu_model(),
            "protocol": "median of timed repetitions after warm-up, torch.set_num_threads(threads), time.perf_counter",
            "legs": legs,
            "sample": (f"same workload ({W}x{H}, {K} kpts, {images} images): " +
                       ", ".join(f"batch {l['batch']} on {l['threads']} threads x {l['reps']} reps ({l['warmup']} warm-up)" +
                                 (" [child process, 45 s limit]" if l["threads"] != cores else "") for l in legs) +
                       f"; value = the fastest leg (batch {best['batch']}, {best['threads']} threads)")}


class PowerSampler:
    """Board power read from the amdgpu hwmon files while the timed region runs (a thread that only reads files: no child process,
    nothing on the GPU).  Evidence for the power-limited regime DESIGN.md describes (round 4: the board sits near 90 % of its cap during
    the step and the clock the chip holds is what gives).  A GPU box shows the hwmon entries of every GPU of its host; which entry
    belongs to this rank's device is not derivable without a child process, so every entry is sampled and the one that draws the most over
    the region -- the busy one -- is reported (at N > 1 every rank reports the maximum over the node's busy GPUs: read it as 'a busy
    GPU', not 'this GPU').  Every field is None where the files are not readable."""

    def __init__(self, index):
        import glob
        import threading

        self._stop = threading.Event()
        self.paths = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*/power1_average")) or \
            sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*/power1_input"))
        self.samples = {p: [] for p in self.paths}
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            for p in self.paths:
                try:
                    self.samples[p].append(int(open(p).read()) / 1e6)
                except (OSError, ValueError):
                    pass
            self._stop.wait(0.05)

    def __enter__(self):
        if self.paths:
            self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.paths:
            self._thread.join(timeout=1.0)

    def summary(self):
        best = None
        for p, v in self.samples.items():
            if v and (best is None or sorted(v)[len(v) // 2] > best[0]):
                best = (sorted(v)[len(v) // 2], max(v), len(v), p)
        if best is None:
            return None
        cap = None
        try:
            cap = int(open(os.path.join(os.path.dirname(best[3]), "power1_cap")).read()) / 1e6
        except (OSError, ValueError):
            pass
        return {"board_power_w_median": round(best[0], 1), "board_power_w_max": round(best[1], 1), "power_cap_w": cap, "samples": best[2],
                "devices_sampled": len(self.paths),
                "source": "amdgpu hwmon power1_average of the busiest visible GPU, sampled every 50 ms over the timed region"}


def launch_ranks(args):
    """Parent of a stand-alone `--gpus N` run: start one rank per device and relay rank 0's JSON line (gluefactory_amd/launch.py).
    Nothing on this path initialises HIP: the devices are counted from the KFD topology in sysfs, the children are fresh processes."""
    from gluefactory_amd.launch import launch_ranks as launch

    raise SystemExit(launch(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))


def family(name):
    """Kernel names that are launches of ONE GPU function share a family: 'lg.attn.self' / 'lg.attn.cross' -> 'lg.attn' (likewise sg. / gs.)."""
    return name.rsplit(".", 1)[0] if name.endswith((".attn.self", ".attn.cross")) else name


def main():
    global K, H, W
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=PAIRS, help="pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="f32: fp32-input MFMA; f16x3: split-fp16 operands on the f16 MFMA (default, fp32-class accuracy)")
    ap.add_argument("--matcher", default="lightglue", choices=["lightglue", "superglue", "gluestick"],
                    help="lightglue: the north-star workload (configs[1]); superglue: configs[3] (use --kpts 2048); "
                         "gluestick: configs[4] with synthetic line segments between detected keypoints (LSD is a CPU dependency)")
    ap.add_argument("--lines", type=int, default=250, help="line segments per image (gluestick)")
    ap.add_argument("--kpts", type=int, default=K, help="keypoints per image")
    ap.add_argument("--size", default=f"{W}x{H}", help="image WxH (default 640x480 = configs[1]; 1024x1024 with --kpts 2048 = the per-GPU share of configs[2])")
    ap.add_argument("--images", default="boat", choices=["boat", "noise"],
                    help="boat: SURVEY 8(d) tiling of the boat pair with circular shifts (default); noise: seeded smooth noise")
    ap.add_argument("--no-f32", action="store_true", help="skip the f32-mode sibling measurement")
    ap.add_argument("--graph", action="store_true", help="replay the step as one HIP graph (gluefactory_amd/graphed.py): for small batches on a busy host, "
                                                         "e.g. --pairs 1; the roofline then comes from the instrumented (eager) pass")
    ap.add_argument("--cpu-leg", default=None, help="internal: 'threads,batch' -- time one leg of the CPU oracle and print it as JSON (no GPU)")
    args = ap.parse_args()
    K = args.kpts
    W, H = (int(v) for v in args.size.lower().split("x"))
    if args.cpu_leg:
        th, bt = (int(v) for v in args.cpu_leg.split(","))
        print(json.dumps(cpu_baseline(matcher=args.matcher, images=args.images, budget_s=20.0, only=(th, bt))), flush=True)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)  # does not return
    # the one child process of a single-GPU run (all-visible-cores CPU leg; only where no cgroup quota defines the CPU share) is started
    # and finished here, before anything below initialises HIP
    extra_leg = None
    if args.gpus == 1 and not args.no_cpu_baseline and args.matcher != "gluestick":
        extra_leg = all_cores_leg(args.matcher, args.images)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a run of another size")
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} has no HIP device {local_rank} (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gluefactory_amd.parallel import all_gather_matches
    from gluefactory_amd.two_view_pipeline import TwoViewPipeline
    from oracle import sp_lg_oracle as O  # seeded weight generators only

    torch.set_grad_enabled(False)
    B = args.pairs
    W_sp = O.make_sp_weights(0)
    W_m = {"lightglue": lambda: O.make_lg_weights(7, final_gain=8.0, match_bias=2.0),
           "superglue": lambda: O.make_sg_weights(9), "gluestick": lambda: O.make_gs_weights(4)}[args.matcher]()

    def build_pipe(precision):
        pp = TwoViewPipeline({
            "extractor": {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": K, "nms_radius": 3,
                          "detection_threshold": 0.0, "remove_borders": 4, "force_num_keypoints": True, "precision": precision},
            "matcher": {"lightglue": {"name": "matchers.lightglue", "filter_threshold": 0.1, "precision": precision},
                        "superglue": {"name": "gluefactory_nonfree.superglue", "filter_threshold": 0.2, "precision": precision,
                                      "return_log_assignment": False},
                        "gluestick": {"name": "matchers.gluestick", "filter_threshold": 0.2, "precision": precision,
                                      "return_log_assignment": False}}[args.matcher],
        }).eval()
        pp.extractor.load_state_dict(W_sp)
        pp.matcher.load_state_dict(W_m)
        return pp

    pipe = build_pipe(args.precision)
    for req in filter(None, os.environ.get("MG_BENCH_DEBUG_REQUEST", "").split(",")):  # A/B timing of kernel variants (e.g. sp.conv1ab_valu)
        pipe.extractor.engine(dev).debug_request(req)
        pipe.matcher.engine(dev).debug_request(req)

    im = make_images(args.images, B, 1000 + rank).to(dev)
    size = torch.tensor([[float(W), float(H)]] * B, device=dev)
    data = {"view0": {"image": im[:B], "image_size": size}, "view1": {"image": im[B:], "image_size": size}}

    if args.matcher == "gluestick":
        # wireframe-style line inputs: the first 2n detected keypoints of every image act as junctions, n segments
        # join random pairs of them (same images every step -> same keypoints, so the segments are built once)
        n = args.lines
        g = torch.Generator().manual_seed(5)
        kp = pipe.extractor({"image": im, "image_size": torch.cat([size, size], 0)})["keypoints"]
        for side, kps in (("0", kp[:B]), ("1", kp[B:])):
            idx = torch.randint(0, 2 * n, (B, n, 2), generator=g)
            idx[:, :, 1] = torch.where(idx[:, :, 1] == idx[:, :, 0], (idx[:, :, 1] + 1) % (2 * n), idx[:, :, 1])
            idx = idx.to(dev)
            data["lines_junc_idx" + side] = idx
            data["lines" + side] = torch.gather(kps, 1, idx.reshape(B, 2 * n)[..., None].expand(-1, -1, 2)).reshape(B, n, 2, 2)
            data["line_scores" + side] = torch.rand(B, n, generator=g).to(dev)

    gather_stats = {}

    graphed = None
    if args.graph:  # the whole step as one hipGraph replay; eager for the instrumented breakdown pass
        if args.matcher != "lightglue":
            raise SystemExit("--graph is verified for the SuperPoint + LightGlue path only (gluefactory_amd/graphed.py)")
        from gluefactory_amd.graphed import GraphedModule
        graphed = GraphedModule(pipe)

    def step(eager=False):
        pred = pipe(data) if (graphed is None or eager) else graphed(data)
        return all_gather_matches(pred, num_pairs=world * B, stats=gather_stats) if world > 1 else pred

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"setup done: {B} pairs/GPU, world {world}")
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    note("warmup done")
    # sanity outside the timed region: every image produced K keypoints, some matches exist
    nvalid = int((out["matches0"] > -1).sum())

    engines = [pipe.extractor.engine(dev), pipe.matcher.engine(dev)]
    no_events = os.environ.get("MG_BENCH_NO_EVENTS") == "1"

    def read_prof():
        acc = {}
        for e in engines:
            for name, ms in e.profile_read(cap=200 * max(args.steps, 1) + 64):
                a = acc.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += 1
        return acc

    # Per-kernel breakdown: a separate, fully instrumented pass OUTSIDE the timed region (an event pair around every launch
    # costs ~10 us of stream time, ~0.7 ms per step over the ~70 launches).  It also tells which kernel dominates.
    nb = max(1, min(args.steps, 3))
    for e in engines:
        e.profile_filter(None)
        e.profile_enable(True)
    for _ in range(nb):
        step(eager=True)
    torch.cuda.synchronize()
    prof_all = read_prof()
    flops_tbl = algorithmic_flops(B)
    # Dominance is judged per GPU FUNCTION, as rocprofv3 --stats lists them: the self and cross launches of a layer are the same
    # attention_f16x3_kernel and count as one kernel ("lg.attn"); every other name of the table is one function (or one instantiation).
    groups = {}
    for k in prof_all:
        if k in flops_tbl:
            groups.setdefault(family(k), []).append(k)
    dom_pre = max(groups, key=lambda g: sum(prof_all[k][0] for k in groups[g]), default=None)
    # timed region: HIP events only around the dominant kernel (the one the roofline is quoted on).  An event pair costs ~10-20 us of
    # stream time (measured round 3: 18 pairs per step around the attention launches took 0.4 ms = 2.5 % off the step), so a kernel that
    # is launched many times per step is sampled in every fourth step of the region only (>= 1 step; the steps are identical).
    dom_launches = sum(prof_all[k][1] for k in groups.get(dom_pre, [])) // nb if dom_pre is not None else 0
    ev_every = 1 if dom_launches <= 2 else 4
    ev_on = not no_events and dom_pre is not None and graphed is None  # (a captured graph holds no event pairs)
    for e in engines:
        e.profile_filter(dom_pre)
        e.profile_enable(ev_on)   # (clears the entries of the breakdown pass)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    power = PowerSampler(local_rank)
    with power:
        t0 = time.perf_counter()
        for i in range(args.steps):
            if ev_on and ev_every > 1:   # pause / resume through the filter (a name no scope has): enable / disable would drop the entries
                for e in engines:
                    e.profile_filter(dom_pre if i % ev_every == 0 else "-")
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    note(f"timed region done: {dt * 1e3 / args.steps:.2f} ms/step")
    prof_dom = read_prof()
    for e in engines:
        e.profile_enable(False)
        e.profile_filter(None)

    # the same workload on the fp32-input MFMA path (the reference's arithmetic), timed the same way in the same run
    f32_mode = None
    if args.precision == "f16x3" and not args.no_f32:
        del out
        pipe32 = build_pipe("f32")
        n32, w32 = max(1, args.steps), max(1, args.warmup)  # the same --steps / --warmup as the headline mode
        for _ in range(w32):
            pipe32(data)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n32):
            p32 = pipe32(data)
            if world > 1:
                all_gather_matches(p32, num_pairs=world * B)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt32 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt32], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt32 = float(t.item())
        f32_mode = {"value": round(world * B * n32 / dt32, 3), "unit": "pairs/s", "dtype": "f32", "steps": n32, "warmup": w32,
                    "ms_per_step": round(dt32 / n32 * 1e3, 3), "peak": PEAK["f32"],
                    "arithmetic": "fp32 storage, fp32-input MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate"}
        note(f"f32 mode: {f32_mode['ms_per_step']:.2f} ms/step")
        del pipe32, p32

    if rank == 0:
        flops = flops_tbl
        prof = prof_all                      # breakdown pass (nb steps, every launch timed)
        total_ms = sum(v[0] for v in prof.values()) or 1.0
        dom = dom_pre if dom_pre is not None else "sp.conv1b"
        members = groups.get(dom, [dom])
        live = [k for k in members if k in prof_dom and prof_dom[k][1]]
        src = prof_dom if len(live) == len(members) else prof   # live: HIP events around this kernel's launches over the timed region
        n_launch = sum(src[k][1] for k in members if k in src)
        avg_ms = (sum(src[k][0] for k in members if k in src) / n_launch) if n_launch else float("nan")
        dom_flops = (sum(flops[k] * src[k][1] for k in members if k in src) / n_launch) if n_launch else float("nan")  # per launch, averaged over the family
        achieved = dom_flops / (avg_ms * 1e-3) / 1e12
        prof_f, flops_f = dict(prof), dict(flops)   # + the families as rows of their own (time share, roofline_top)
        for g, ks in groups.items():
            if len(ks) > 1:
                prof_f[g] = [sum(prof_all[k][0] for k in ks), sum(prof_all[k][1] for k in ks)]
                flops_f[g] = sum(flops_tbl[k] * prof_all[k][1] for k in ks) / max(1, sum(prof_all[k][1] for k in ks))
        kernels = {k: {"ms_per_step": round(v[0] / nb, 4), "launches_per_step": v[1] // nb,
                       "tflops": round(flops[k] * (v[1] / nb) / (v[0] / nb * 1e-3) / 1e12, 2) if k in flops else None}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        # HBM bytes per launch of the dominant kernel: from the committed rocprofv3 PMC passes of this same
        # command (profiles/summarize_pmc.py; FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), not live
        traffic, traffic_src = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.precision}_hbm_traffic.json")))
        tpath = cands[-1] if cands else ""  # the newest committed PMC summary for this arithmetic mode
        if tpath and B == PAIRS and args.matcher == "lightglue" and K == 1024 and (W, H) == (640, 480):
            tj = json.load(open(tpath))
            if dom in tj:
                traffic, traffic_src = tj[dom]["hbm_bytes_per_launch"], os.path.relpath(tpath, ROOT)
        res = {
            "metric": ("image-pairs/sec SP+LightGlue (VGA, 1024 kpts) at 1/2/4/8 MI355X" if args.matcher == "lightglue" and K == 1024 and (W, H) == (640, 480)
                       else f"image-pairs/sec SP+{args.matcher} (VGA, {K} kpts)"),
            "value": round(world * B * args.steps / dt, 3),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": (f"SuperPoint + LightGlue, {W}x{H}, {K} kpts, batch={B} pairs per GPU (BASELINE.json configs[{1 if (W, H, K) == (640, 480, 1024) else 2}])"
                                    if args.matcher == "lightglue" else
                                    f"SuperPoint + SuperGlue, {W}x{H}, {K} kpts, batch={B} pairs per GPU (BASELINE.json configs[3])"
                                    if args.matcher == "superglue" else
                                    f"SuperPoint + GlueStick, {W}x{H}, {K} kpts + {args.lines} synthetic segments, batch={B} pairs per GPU "
                                    "(BASELINE.json configs[4]; LSD itself is a CPU dependency and not timed)"),
                       "pairs_per_gpu": B, "pairs_per_rank": [B] * world, "global_pairs": B * world, "parallelism": f"dp{world}",
                       "images": args.images, "matches_per_step_rank0": nvalid, "hip_graph_replay": bool(args.graph),
                       "all_gather": ({"collective": "all_gather_into_tensor (RCCL)", "calls_per_step": 1,
                                       "bytes_sent_per_rank": gather_stats.get("bytes_sent"), "bytes_received_per_rank": gather_stats.get("bytes_received")}
                                      if world > 1 else None),
                       "arithmetic": ("fp32 storage, fp32-input MFMA, fp32 accumulate" if args.precision == "f32" else
                                      "fp32-equivalent split-fp16 storage (hi+lo), f16 MFMA x3 products, fp32 accumulate")},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK[args.precision], "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK[args.precision], 4), "traffic": traffic, "traffic_source": traffic_src,
                         "mfma_products_per_flop": 1 if args.precision == "f32" else 3,
                         "executed_frac": round(achieved * (1 if args.precision == "f32" else 3) / PEAK[args.precision], 4),
                         "sustained_peak_measured": SUSTAINED[args.precision],
                         "executed_frac_of_sustained": (round(achieved * 3 / SUSTAINED[args.precision], 4) if SUSTAINED[args.precision] else None),
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_flops_per_launch": dom_flops,
                         "avg_launch_source": (f"HIP events around {dom} ({' + '.join(members)}) in " +
                                               (f"each of the {args.steps} timed steps" if ev_every == 1 else f"every {ev_every}th of the {args.steps} timed steps")
                                               if src is prof_dom else f"instrumented pass of {nb} steps"),
                         "launches_per_step": dom_launches,
                         "kernel_time_share": round(prof_f[dom][0] / total_ms, 4) if dom in prof_f else None},
            "kernels": kernels,
            "kernels_source": f"separate fully instrumented pass of {nb} steps before the timed region (an event pair per launch costs ~10 us of stream time; "
                              "the timed region carries the dominant kernel's pair only)",
        }
        # the four largest GPU functions with the same figures, from the breakdown pass: three of them are within 15 % of each other
        # (round 3: attention self + cross 3.77 ms, fused FFN 3.38, conv1ab 3.24-3.44 ms per step), so a reader sees all of them in every line
        mult = 1 if args.precision == "f32" else 3
        res["roofline_top"] = [
            {"kernel": k, "ms_per_step": round(prof_f[k][0] / nb, 4), "launches_per_step": prof_f[k][1] // nb, "avg_launch_ms": round(prof_f[k][0] / prof_f[k][1], 4),
             "achieved": round(flops_f[k] / (prof_f[k][0] / prof_f[k][1] * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
             "frac": round(flops_f[k] / (prof_f[k][0] / prof_f[k][1] * 1e-3) / 1e12 / PEAK[args.precision], 4),
             "executed_frac": round(flops_f[k] / (prof_f[k][0] / prof_f[k][1] * 1e-3) / 1e12 * mult / PEAK[args.precision], 4),
             "kernel_time_share": round(prof_f[k][0] / total_ms, 4)}
            for k in sorted((k for k in prof_f if k in flops_f and not (family(k) != k and family(k) in prof_f)), key=lambda k: -prof_f[k][0])[:4]]
        att = [k for k in prof if ".attn." in k and k in flops]
        if att:  # the attention contractions (QK^T + PV) on their own: north_star asks for their MFMA utilisation
            a_ms = sum(prof[k][0] for k in att)
            a_fl = sum(flops[k] * prof[k][1] for k in att)
            a_tf = a_fl / (a_ms * 1e-3) / 1e12
            mult = 1 if args.precision == "f32" else 3
            res["attention_mfma"] = {"kernels": sorted(att), "achieved_algorithmic": round(a_tf, 2), "executed": round(a_tf * mult, 2),
                                     "peak": PEAK[args.precision], "unit": "TFLOP/s", "executed_frac": round(a_tf * mult / PEAK[args.precision], 4),
                                     "executed_frac_of_sustained": (round(a_tf * mult / SUSTAINED[args.precision], 4) if SUSTAINED[args.precision] else None),
                                     "ms_per_step": round(a_ms / nb, 4)}
        if "sg.sinkhorn" in prof:  # configs[3]: the log-domain Sinkhorn against its byte floor (one sweep of the couplings per iteration)
            it, ms = 50, prof["sg.sinkhorn"][0] / nb
            floor = float(B) * (K + 1) * (K + 1) * 4 * it
            res["sinkhorn"] = {"ms_per_step": round(ms, 4), "iterations": it, "floor_bytes_per_step": floor,
                               "floor_definition": "B x (K+1)^2 x 4 B read once per iteration (superglue.py:173-201: 50 iterations, two half-steps each)",
                               "achieved_GBps_vs_floor_bytes": round(floor / (ms * 1e-3) / 1e9, 1), "hbm_peak_GBps": 8000.0}
        if power.summary() is not None:
            res["power"] = power.summary()
        if f32_mode is not None:
            res["f32_mode"] = f32_mode
        if world == 1 and not args.no_cpu_baseline:
            note("timing the CPU oracle (B = 1 and B = 8 on the core share, + all visible cores when no cgroup quota defines the share; 3 warm-up + up to 10 repetitions each, 1-2 min)")
            cb = cpu_baseline(matcher=args.matcher, images=args.images, note=note, extra_leg=extra_leg)
            if cb is not None:
                res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

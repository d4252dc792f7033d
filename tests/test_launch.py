"""The stand-alone N-rank launcher of `bench.py --gpus N` (gluefactory_amd/launch.py), driven on the CPU with stub children: the one
path of the bench that has never run on hardware must not fail for avoidable reasons.  Plus the distributed leg itself at the
configs[2] split (256 pairs over 8 ranks) and an uneven split, over gloo."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def _run_launcher(tmp_path, n, child, n_devices):
    code = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'glue-factory_amd')!r})
from gluefactory_amd.launch import launch_ranks
sys.exit(launch_ranks({n}, {child!r}, n_devices={n_devices!r}))
"""
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)


@pytest.mark.parametrize("world", [2, 3])
def test_launcher_relays_one_json_line_from_rank0(tmp_path, world):
    child = _stub(tmp_path, """
        import json, os, socket
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        # a real rendezvous on the port the launcher chose: every rank reaches the store
        from torch.distributed import TCPStore
        st = TCPStore("127.0.0.1", int(os.environ["MASTER_PORT"]), w, r == 0)
        st.add("ready", 1)
        st.wait(["ready"])
        import time
        while int(st.add("ready", 0)) < w:
            time.sleep(0.01)
        st.add("done", 1)
        if r == 0:
            while int(st.add("done", 0)) < w:
                time.sleep(0.01)
            print(json.dumps({"n_gpus": w, "rank": r}), flush=True)
    """)
    res = _run_launcher(tmp_path, world, child, world)
    assert res.returncode == 0, res.stderr
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": world, "rank": 0}


def test_launcher_stops_siblings_when_a_rank_dies(tmp_path):
    child = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)  # a rank waiting in a collective that will never complete
    """)
    import time

    t0 = time.time()
    res = _run_launcher(tmp_path, 3, child, 3)
    assert res.returncode == 7 and time.time() - t0 < 60
    assert res.stdout.strip() == "" and "rank 1 exited with 7" in res.stderr


def test_launcher_kills_a_rank_that_ignores_sigterm(tmp_path):
    """A sibling stuck where SIGTERM does not reach it (a collective, a GPU wait) must not keep the launcher alive: after the grace
    period it is killed and the first failing rank's exit code is still what the launcher leaves with."""
    child = _stub(tmp_path, """
        import os, signal, sys, time
        if int(os.environ["RANK"]) == 1:
            time.sleep(0.5)
            sys.exit(7)
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(600)
    """)
    code = f"""
import sys, time
sys.path.insert(0, {os.path.join(ROOT, 'glue-factory_amd')!r})
from gluefactory_amd.launch import launch_ranks
t0 = time.time()
rc = launch_ranks(2, {child!r}, n_devices=2, grace_s=1.0)
print("elapsed", time.time() - t0)
sys.exit(rc)
"""
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert res.returncode == 7, (res.returncode, res.stderr)
    assert "ignored SIGTERM" in res.stderr
    assert float(res.stdout.split()[-1]) < 30


def test_launcher_refuses_when_devices_are_missing(tmp_path):
    child = _stub(tmp_path, "print('must not start')")
    res = _run_launcher(tmp_path, 4, child, 2)
    assert res.returncode == 2 and "must not start" not in res.stdout and "refusing" in res.stderr
    # unknown device count (no KFD topology, as in this container): the children decide
    res = _run_launcher(tmp_path, 2, _stub(tmp_path, "import sys; sys.exit(0)"), None)
    assert res.returncode == 0


def test_count_gpus_reads_the_kfd_topology(tmp_path, monkeypatch):
    from gluefactory_amd.launch import count_gpus

    for i, simd in enumerate((0, 1024, 1024, 1024)):  # node 0 = the CPU
        d = tmp_path / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert count_gpus(str(tmp_path / "nodes")) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert count_gpus(str(tmp_path / "nodes")) == 2
    assert count_gpus(str(tmp_path / "nope")) is None


def test_bench_refuses_a_world_size_other_than_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode != 0 and res.stdout.strip() == "" and "WORLD_SIZE=3" in res.stderr


def test_bench_parent_never_calls_torch_cuda():
    """The launcher branch of bench.py: no torch.cuda call (a HIP-initialised parent must not spawn / exec on this pool)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks(args):"):src.index("def main():")]
    assert "torch.cuda" not in body
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args)") < main.index("torch.cuda")
    # the single-GPU run's only child process (all-visible-cores CPU leg) is also started before the first torch.cuda call, and the
    # CPU-baseline code that runs after the GPU work spawns nothing
    assert main.index("all_cores_leg(") < main.index("torch.cuda")
    cb = src[src.index("def cpu_baseline("):src.index("def launch_ranks(args):")]
    assert "subprocess" not in cb and "Popen" not in cb
    assert "torch" not in open(os.path.join(ROOT, "glue-factory_amd", "gluefactory_amd", "launch.py")).read().split('"""', 2)[2]


# ---------------------------------------------------------------------------------------------- the data path of an N-rank step, over gloo
def _fake_forward(data):
    """A stand-in for the per-rank SuperPoint + LightGlue forward with the real output shapes and dtypes: every output row is a
    deterministic function of its own pair only, as in the real path (pairs are independent)."""
    a, b = data["view0"]["image"], data["view1"]["image"]
    n, k = a.shape[0], 16
    s = (a.flatten(1).sum(1, keepdim=True) * 31 + b.flatten(1).sum(1, keepdim=True) * 17 + torch.arange(k)[None]).double()
    m0 = (s % (k + 1)).long() - 1
    return {"matches0": m0, "matches1": m0.flip(1), "matching_scores0": (s % 97 / 97).float(), "matching_scores1": (s % 89 / 89).float(),
            "keypoints0": torch.stack([(s % 640).float(), (s % 480).float()], -1)}


def _step_worker(rank, world, port, n_pairs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gluefactory_amd.parallel import all_gather_matches, shard_batch, shard_range

    g = torch.Generator().manual_seed(11)
    full = {"view0": {"image": torch.randint(0, 255, (n_pairs, 1, 4, 4), generator=g).float(), "image_size": torch.ones(n_pairs, 2)},
            "view1": {"image": torch.randint(0, 255, (n_pairs, 1, 4, 4), generator=g).float(), "image_size": torch.ones(n_pairs, 2)}}
    want = _fake_forward(full)
    local = shard_batch(full, rank, world)
    lo, hi = shard_range(n_pairs, rank, world)
    ok = local["view0"]["image"].shape[0] == hi - lo
    stats = {}
    got = all_gather_matches(_fake_forward(local), num_pairs=n_pairs, with_keypoints=True, stats=stats)
    ok = ok and all(got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]) for k in want)
    ok = ok and stats["bytes_received"] == world * stats["bytes_sent"]
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_pairs", [(8, 256), (8, 250), (4, 30)])
def test_shard_forward_gather_step(world, n_pairs):
    """configs[2]: 256 pairs over 8 ranks (32 each); 250 over 8 (shards of 32 and 31) and 30 over 4 (8, 8, 7, 7) exercise the padding.
    shard_batch -> per-rank forward -> one packed all-gather reproduces the single-process result on every rank."""
    os.environ["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "glue-factory_amd"), os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")])
    old = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "1"  # eight ranks on eight cores; set for the children only (this process keeps its thread count:
    try:                                   # the oracle-vs-golden tests compare fp32 sums whose order depends on it)
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_step_worker, args=(world, _free_port(), n_pairs, out), nprocs=world, join=True)
    finally:
        if old is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = old
    assert all(out[r] for r in range(world)) and len(out) == world

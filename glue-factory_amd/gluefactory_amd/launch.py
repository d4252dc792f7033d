"""Stand-alone launcher of an N-rank single-node run: one child process per GPU, before anything in the parent touches HIP.

`bench.py --gpus N` uses it when no launcher (torch.distributed.run) has set WORLD_SIZE -- the pattern of
gluefactory/train.py:699-701 (`torch.multiprocessing.spawn(main_worker, nprocs=n_gpus)`), with plain child processes instead of a
fork of a process that might own a GPU context.  The parent imports nothing that initialises the runtime: the devices are counted
from the KFD topology in sysfs (a node with SIMDs is a GPU), honouring HIP_/ROCR_VISIBLE_DEVICES lists; when sysfs cannot be read
the count is unknown and the children themselves refuse (a rank without its device exits non-zero).

Contract (tests/test_launch.py drives it with a stub child):
  * every child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT (a free loopback port) and inherits
    stdout / stderr, so rank 0's single JSON line is the parent's output;
  * the parent's exit code is the first non-zero child exit code; when one rank dies the others are terminated (they would wait in a
    collective forever) and killed if they ignore that;
  * fewer visible devices than ranks asked for: refuse before starting anything.
"""
import glob
import os
import socket
import subprocess
import sys
import time


def count_gpus(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process would see, without initialising HIP: KFD nodes with simd_count > 0, cut to the visible-devices list if one is
    set.  None when the topology cannot be read."""
    n = 0
    found = False
    for path in glob.glob(os.path.join(topology, "*", "properties")):
        try:
            props = dict(line.split()[:2] for line in open(path) if len(line.split()) >= 2)
        except OSError:
            continue
        found = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    if not found:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n_ranks, cmd, n_devices="auto", poll_s=0.05, grace_s=5.0):
    """Start `cmd` (argv list) once per rank and wait; returns the exit code to leave with (0 = all ranks succeeded)."""
    if n_devices == "auto":
        n_devices = count_gpus()
    if n_devices is not None and n_devices < n_ranks:
        print(f"launch: {n_ranks} ranks were asked for but {n_devices} GPU(s) are visible; refusing to report a smaller run", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    kill_at = None  # set at the first failure: ranks still alive `grace_s` later ignored SIGTERM (stuck in a collective / a GPU wait)
    try:
        pending = set(range(n_ranks))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"launch: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
                    kill_at = time.time() + grace_s
            if kill_at is not None and pending and time.time() >= kill_at:
                for q in sorted(pending):
                    print(f"launch: rank {q} ignored SIGTERM for {grace_s:g} s; killing it", file=sys.stderr)
                    procs[q].kill()
                kill_at = float("inf")  # SIGKILL cannot be ignored: the loop ends as soon as the kernel has reaped them
            time.sleep(poll_s)
    finally:
        t_end = time.time() + grace_s
        for pr in procs:
            while pr.poll() is None and time.time() < t_end:
                time.sleep(poll_s)
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    return rc

"""Start N rank processes of one node from a parent that never touches the GPU.

The reference is single-device (main_yelp.py:125: one `torch.cuda.set_device`); a job over several GPUs here is one
process per GPU.  `bench.py --gpus N` and `main_yelp.py --gpus N` call spawn_ranks() BEFORE anything initialises HIP in
the calling process (importing torch does not; `torch.cuda.device_count()` does not on this image): every child is a
fresh interpreter with

    RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT
    HSA_ENABLE_IPC_MODE_LEGACY=0      the host driver only does dmabuf IPC: without it hipIpcGetMemHandle fails and the
                                      one-shot peer exchange (and RCCL's own IPC) cannot map another rank's memory
    SML_LAUNCHED=1                    "this process is a rank": the child runs the workload instead of spawning again

Rank 0's stdout is the job's stdout (bench.py: the ONE JSON line); the other ranks' stdout goes to stderr.  The first
rank that exits non-zero ends the job: the others are terminated (by PID) and the launcher returns that code.

one_device=True (test mode, a 1-GPU box): every rank gets LOCAL_RANK=0 and SML_ONE_DEVICE=1 -- the ranks share device 0
as separate processes (separate HIP contexts and queues, hipIpc mappings between them), with gloo carrying
torch.distributed because RCCL refuses two ranks on one device.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, one_device=False, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK="0" if one_device else str(rank), WORLD_SIZE=str(world),
               LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0", SML_LAUNCHED="1")
    if one_device:
        env["SML_ONE_DEVICE"] = "1"
    else:
        env.pop("SML_ONE_DEVICE", None)
    return env


def is_rank_process():
    """True inside a process that spawn_ranks (or torchrun) started as a rank."""
    return os.environ.get("SML_LAUNCHED") == "1" or int(os.environ.get("WORLD_SIZE", "1")) > 1


def prepare_rank_env():
    """For rank processes started by something else (torchrun): the IPC mode must be in the environment before the first
    HIP call of the process (the runtime reads it when it initialises)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def one_device():
    return os.environ.get("SML_ONE_DEVICE") == "1"


def backend():
    """torch.distributed backend of a rank process: RCCL ("nccl") one process per GPU; gloo when the ranks share a device."""
    return "gloo" if one_device() else "nccl"


def spawn_ranks(argv, world, one_device=False, timeout=None, echo_stdout=True, env=None):
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) as `world` rank processes.  Returns
    (exit code, rank 0's stdout text).  Never imports or calls anything that initialises the GPU."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, one_device, env),
                                      stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))
    chunks = [[] for _ in procs]

    def pump(r):
        for line in procs[r].stdout:
            chunks[r].append(line)
            if r != 0:
                sys.stderr.write("[rank %d] %s" % (r, line))
                sys.stderr.flush()
    threads = [threading.Thread(target=pump, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    t0 = time.time()
    code = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = rc
                sys.stderr.write("[sml_amd.launch] rank %d exited with code %d: stopping the other ranks\n" % (r, rc))
        if code != 0 or (timeout is not None and time.time() - t0 > timeout):
            if code == 0:
                code = 124
                sys.stderr.write("[sml_amd.launch] time-out after %.0f s: stopping the ranks\n" % timeout)
            for r in sorted(live):
                procs[r].terminate()             # (exact PIDs this launcher started)
            deadline = time.time() + 10.0
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
            break
        if live:
            time.sleep(0.05)
    for t in threads:
        t.join(timeout=5.0)
    out = "".join(chunks[0])
    if echo_stdout and out:
        sys.stdout.write(out)
        sys.stdout.flush()
    return code, out

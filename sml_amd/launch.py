"""Start N rank processes of one node from a parent that never touches the GPU.

The reference is single-device (main_yelp.py:125: one `torch.cuda.set_device`); a job over several GPUs here is one
process per GPU.  `bench.py --gpus N` and `main_yelp.py --gpus N` call spawn_ranks() BEFORE anything initialises HIP in
the calling process (importing torch does not; the GPU count comes from sysfs, `visible_gpus()`): every child is a
fresh interpreter with

    RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT
    HSA_ENABLE_IPC_MODE_LEGACY=0      the host driver only does dmabuf IPC: without it hipIpcGetMemHandle fails and the
                                      one-shot peer exchange (and RCCL's own IPC) cannot map another rank's memory
    SML_LAUNCHED=1                    "this process is a rank": the child runs the workload instead of spawning again

Rank 0's stdout is the job's stdout (bench.py: the ONE JSON line; main_yelp.py: the training log, relayed line by line as
it arrives); the other ranks' stdout goes to stderr.  The first rank that exits non-zero ends the job: the others are
terminated (by the process groups this launcher created) and the launcher returns that code.  The ranks never outlive the
launcher: try/finally + SIGTERM / SIGINT / SIGHUP handlers stop them, PR_SET_PDEATHSIG covers a SIGKILLed launcher, and
`timeout` (bench.py --job-timeout, main_yelp.py --job_timeout, SML_JOB_TIMEOUT_S) bounds a job whose ranks hang.

one_device=True (test mode, a 1-GPU box): every rank gets LOCAL_RANK=0 and SML_ONE_DEVICE=1 -- the ranks share device 0
as separate processes (separate HIP contexts and queues, hipIpc mappings between them), with gloo carrying
torch.distributed because RCCL refuses two ranks on one device.
"""
import os
import signal
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


def visible_gpus():
    """GPU count of this node WITHOUT a HIP / HSA call: the KFD topology in sysfs (nodes with SIMDs are GPUs, narrowed
    by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES -- torch on ROCm honours the last one and
    cli.py sets it -- when they are plain lists; a variable that is set and EMPTY hides every device).  None when the
    topology cannot be read.
    (`torch.cuda.device_count()` stays off HIP only while its amdsmi path works; its fall-back is hipGetDeviceCount, which
    initialises the runtime in the parent of the rank processes -- ADVICE r4.)"""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    if k == "simd_count":
                        n += int(v) > 0
                        break
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def job_timeout(flag_value=None, default=None):
    """Seconds a launched job may run: the command line's value, else SML_JOB_TIMEOUT_S, else `default`; <= 0: no limit."""
    v = flag_value if flag_value is not None else os.environ.get("SML_JOB_TIMEOUT_S")
    if v is None or v == "":
        v = default
    if v is None:
        return None
    v = float(v)
    return v if v > 0 else None


# PR_SET_PDEATHSIG for the rank processes: resolved at import, so that the hook between fork and exec is ONE C call (no import,
# no allocation-heavy Python in a child forked from a process that may have threads -- ADVICE r5); the ranks' own session
# (= process group, so the launcher can signal a rank AND whatever it started with one killpg) comes from
# Popen(start_new_session=True), which is setsid() done by the interpreter's C code
try:
    import ctypes as _ctypes
    _prctl = _ctypes.CDLL(None, use_errno=True).prctl
    _prctl.argtypes = [_ctypes.c_int, _ctypes.c_ulong, _ctypes.c_ulong, _ctypes.c_ulong, _ctypes.c_ulong]
    _prctl.restype = _ctypes.c_int
except Exception:                                  # pragma: no cover  (no libc prctl: the launcher's own clean-up still runs)
    _prctl = None
_SIGKILL = int(signal.SIGKILL)


def _rank_preexec():
    # in the child, before exec: SIGKILL from the kernel if the launcher's thread dies first (a launcher that was
    # SIGKILLed cannot clean up: PR_SET_PDEATHSIG = 1)
    if _prctl is not None:
        _prctl(1, _SIGKILL, 0, 0, 0)


def _stop(procs, grace=10.0):
    """Terminate, then kill, every rank still alive -- by the exact process groups this launcher created."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError):
            pass
    deadline = time.time() + grace
    for p in live:
        try:
            p.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def spawn_ranks(argv, world, one_device=False, timeout=None, echo_stdout=True, env=None):
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) as `world` rank processes.  Returns
    (exit code, rank 0's stdout text).  Never imports or calls anything that initialises the GPU.

    Rank 0's stdout is relayed LINE BY LINE as it arrives (a training log shows while the job runs) and collected for the
    return value.  Whatever ends this call -- a rank failing, the time-out (exit code 124), SIGTERM / SIGINT / SIGHUP to the
    launcher, an exception -- every rank's process group is terminated, then killed; a launcher that is SIGKILLed takes its
    ranks with it through PR_SET_PDEATHSIG."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    chunks = [[] for _ in range(world)]
    threads = []
    old_handlers = {}

    class _Signalled(Exception):
        pass

    cleaning = []                                  # non-empty once the clean-up runs: further signals are recorded, not raised

    def on_signal(signum, frame):
        if cleaning:
            cleaning.append(signum)
            return
        raise _Signalled(signum)

    def pump(r):
        for line in procs[r].stdout:
            chunks[r].append(line)
            if r != 0:
                sys.stderr.write("[rank %d] %s" % (r, line))
                sys.stderr.flush()
            elif echo_stdout:
                sys.stdout.write(line)
                sys.stdout.flush()
    code = 0
    try:
        if threading.current_thread() is threading.main_thread():
            for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                old_handlers[sg] = signal.signal(sg, on_signal)
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port, one_device, env), preexec_fn=_rank_preexec,
                                          start_new_session=True, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))
        threads = [threading.Thread(target=pump, args=(r,), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        t0 = time.time()
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
                break
            if live:
                time.sleep(0.05)
    except _Signalled as e:
        signum = int(e.args[0])
        sys.stderr.write("[sml_amd.launch] signal %d: stopping the ranks\n" % signum)
        code = 128 + signum
    finally:
        cleaning.append(0)                         # a second SIGTERM / SIGINT must not abort the kill loop below
        _stop(procs)                               # (no-op for ranks that have exited)
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
        for t in threads:
            t.join(timeout=5.0)
    return code, "".join(chunks[0])

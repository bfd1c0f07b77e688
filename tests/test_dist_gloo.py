"""N > 1 path on CPU: two gloo ranks run the product's distributed logic (sml_amd.dist:
user sharding, global item-occurrence lists, item-gradient all-gather, theta-gradient
all-reduce, loss scaling) against a one-process run of the same global batches."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import make_mf, make_transfer, quiet  # noqa: E402
from _cpu_engine import CpuDistEngine, CpuEngine    # noqa: E402

U, I, D, B = 40, 30, 32, 16
LR, L2, TR_LR, TR_WD = 0.01, 1e-6, 1e-3, 1e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(n, seed=0):
    """Per-rank triples (users inside the rank's shard) and shared initial state."""
    from sml_amd import dist as SD
    g = torch.Generator().manual_seed(seed)
    tri = []
    for r in range(2):
        lo, hi = SD.user_range(U, 2, r)
        u = torch.randint(lo, hi, (n,), generator=g)
        u[:5] = lo                                   # duplicates inside a batch
        i = torch.randint(0, I, (n,), generator=g)
        j = torch.randint(0, I, (n,), generator=g)
        j[3] = i[3]
        tri.append(torch.stack([u, i, j], 1))
    wu = torch.randn(U, D, generator=g) * 0.3
    wi = torch.randn(I, D, generator=g) * 0.3
    torch.manual_seed(seed + 1)
    with quiet():
        net = make_transfer(D)
    return tri, wu, wi, {k: v.clone() for k, v in net.state_dict().items()}


def _run_rank(rank, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from sml_amd import dist as SD
        from sml_amd.period import PeriodState
        tri, wu, wi, sd = _data(n)
        mf = make_mf(U, I, D, wu.numpy(), wi.numpy())
        with quiet():
            net = make_transfer(D)
        if rank == 0:
            net.load_state_dict(sd)                   # rank 1 starts from different values: attach must sync
        else:
            with torch.no_grad():
                mf.item_laten.weight.add_(1.0)
        st = PeriodState(mf, net)
        eng = CpuDistEngine(d=D)
        SD.attach(eng, st, dist)
        lu, li = wu * 0.9, wi * 0.9
        l_mf = eng.mf_stage_epoch(mf, net, lu, li, tri[rank], B, LR, L2)
        hu, hi = mf.user_laten.weight.detach().clone(), mf.item_laten.weight.detach().clone()
        l_tr = eng.tr_stage_epoch(net, lu, li, hu, hi, tri[rank], B, TR_LR, TR_WD)
        torch.save(dict(wu=mf.user_laten.weight.detach(), wi=mf.item_laten.weight.detach(),
                        theta={k: v.clone() for k, v in net.state_dict().items()}, l_mf=l_mf, l_tr=l_tr),
                   os.path.join(out, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 60])        # 60: ragged last batch (12 per rank)
def test_two_ranks_equal_one_rank_global_batches(tmp_path, n):
    mp.spawn(_run_rank, args=(_free_port(), n, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"), weights_only=False)
    # replicas: item table and theta identical on both ranks
    assert torch.equal(r0["wi"], r1["wi"])
    for k in r0["theta"]:
        assert torch.equal(r0["theta"][k], r1["theta"][k]), k
    # one process, global batches = [rank0 batch b ; rank1 batch b], batch size 2B
    from sml_amd import dist as SD
    tri, wu, wi, sd = _data(n)
    glob = torch.cat([torch.cat([tri[0][b:b + B], tri[1][b:b + B]]) for b in range(0, n, B)])
    mf = make_mf(U, I, D, wu.numpy(), wi.numpy())
    with quiet():
        net = make_transfer(D)
    net.load_state_dict(sd)
    eng = CpuEngine(d=D)
    lu, li = wu * 0.9, wi * 0.9
    l_mf = eng.mf_stage_epoch(mf, net, lu, li, glob, 2 * B, LR, L2)
    hu, hi = mf.user_laten.weight.detach().clone(), mf.item_laten.weight.detach().clone()
    # the TR stage of each rank saw ITS user rows updated and the shared item table
    lo1, hi1 = SD.user_range(U, 2, 1)
    wu_dist = r0["wu"].clone()
    wu_dist[lo1:hi1] = r1["wu"][lo1:hi1]
    np.testing.assert_allclose(wu_dist.numpy(), hu.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(r0["wi"].numpy(), hi.numpy(), rtol=2e-4, atol=2e-6)
    l_tr = eng.tr_stage_epoch(net, lu, li, hu, hi, glob, 2 * B, TR_LR, TR_WD)
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(r0["theta"][k].numpy(), v.numpy(), rtol=5e-4, atol=5e-6)
    # losses: the global batch loss is the sum of the ranks' scaled parts
    np.testing.assert_allclose(r0["l_mf"] + r1["l_mf"], l_mf, rtol=1e-5)
    np.testing.assert_allclose(r0["l_tr"] + r1["l_tr"], l_tr, rtol=1e-5)


def test_user_range_and_item_lists():
    from sml_amd import dist as SD
    assert [SD.user_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert SD.owner_of(torch.tensor([0, 2, 3, 9]), 10, 4).tolist() == [0, 0, 1, 3]
    tri = torch.tensor([[[0, 5, 7], [1, 5, 2], [2, 9, 5]], [[3, 7, 5], [4, 1, 1], [5, 5, 0]]])   # world 2, n 3
    keys, vals = SD.global_item_lists(tri, batch=2)
    rows, bat = (keys & 0xffffffff).tolist(), (keys >> 32).tolist()
    # batch-major (batch 0's 8 occurrences, then batch 1's 4); inside a batch in (rank, positive / negative, element) order and NOT
    # sorted: the library builds the run lists (round 5: no library sort on the product path)
    assert bat == sorted(bat) and bat.count(0) == 8 and bat.count(1) == 4
    assert rows[:8] == [5, 5, 7, 2, 7, 1, 5, 1] and rows[8:] == [9, 5, 5, 0]
    # value = slot in [world][2*batch]: rank q positives q*4 + t, negatives q*4 + B_b + t
    lookup = {}
    for q in range(2):
        for e in range(3):
            b, t = divmod(e, 2)
            Bb = min(2, 3 - 2 * b)
            lookup[(b, q * 4 + t)] = int(tri[q, e, 1])
            lookup[(b, q * 4 + Bb + t)] = int(tri[q, e, 2])
    for k, v, b in zip(rows, vals.tolist(), bat):
        assert lookup[(b, v)] == k


# ----------------------------------------------------------------------------- the real driver under 2 ranks
def _driver_rank(rank, world, port, root, ck, ttype, out):
    """main_yelp.py's program (sml_amd.cli: seeds, transfer_data, meta_train.run) on a six-period dataset, under
    `world` gloo ranks (world 1: the plain single-process driver), with the oracle-based engine doubles."""
    import contextlib
    import io
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from sml_amd import cli, datasets, driver
    from sml_amd.mf import MFbasemode
    from test_host_logic import _cpu_mf_test
    driver._default_device = lambda: torch.device("cpu")
    driver._make_engine = (lambda dev, d, mb: CpuDistEngine(dev, d, mb)) if world > 1 else (lambda dev, d, mb: CpuEngine(dev, d, mb))
    MFbasemode.test = _cpu_mf_test
    d = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = dist
    try:
        args = cli.get_parse("yelp").parse_args(["--data_path", root, "--pre_model", ck, "--laten", "32", "--multi_num", "2",
                                                "--numworkers", "0", "--MF_batch_size", "64", "--TR_batch_size", "32",
                                                "--transfer_type", ttype])
        torch.manual_seed(args.seed)
        np.random.seed(args.seed + 2)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            sets = datasets.transfer_data(args, path=root, datasetname="yelp", file_path_list=[str(i) for i in range(6)],
                                          test_list=[str(j) for j in range(3, 6)], validation_list=None, online_train_time=1,
                                          online_test_time=3)
            meta = driver.meta_train(args, sets, sets.user_number, sets.item_number, args.laten, dist=d)
            meta.run(args)
        torch.save(dict(log=buf.getvalue(), recall=[float(r) for r in meta.recall], wi=meta.MFbase.item_laten.weight.detach().clone(),
                        wu=meta.MFbase.user_laten.weight.detach().clone(),
                        theta={k: v.clone() for k, v in meta.transfer.state_dict().items()}),
                   os.path.join(out, "w%d_rank%d.pt" % (world, rank)))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_driver_under_four_ranks_prints_the_single_process_run(tmp_path):
    """The same with FOUR ranks (round 5: nothing had run the host logic -- routing by owner, per-rank batch counts, the job-wide
    item lists, loss scaling, routed evaluation -- at a world size above two before the first 8-GPU job would): 75 users per rank,
    many local batches empty."""
    _driver_world_check(tmp_path, "conv_com", 4)


@pytest.mark.parametrize("ttype", ["conv_com", "conv"])        # BCE (a mean over the split batch) and BPR (a sum)
def test_driver_under_two_ranks_prints_the_single_process_run(tmp_path, ttype):
    _driver_world_check(tmp_path, ttype, 2)


def _driver_world_check(tmp_path, ttype, world):
    """`torchrun --nproc-per-node 2 main_yelp.py` in miniature: users row-sharded by owner (each rank keeps its rows
    only), every global batch split by owner -- unequal, sometimes empty local batches --, item-gradient exchange,
    theta all-reduce, routed evaluation.  The two ranks must print what one process prints: same lines, losses to
    1e-4, recall within 2 rank flips of the 160-row sets; item replicas and theta bit-identical between the ranks."""
    import re
    from sml_amd import synth
    from sml_amd.mf import MFbasemode
    root = str(tmp_path) + "/"
    U, I = 300, 120
    synth.write_dataset(root, "yelp", n_periods=6, n_inter=160, n_user=U, n_item=I, neg=49, a_user=0.8, a_item=0.8, seed=77)
    torch.manual_seed(9)
    mf = MFbasemode(U, I, 32)
    with torch.no_grad():
        mf.user_laten.weight.mul_(0.3)
        mf.item_laten.weight.mul_(0.3)
    ck = root + "init.pkl"
    torch.save(mf, ck)
    out = str(tmp_path)
    mp.spawn(_driver_rank, args=(1, 0, root, ck, ttype, out), nprocs=1, join=True)     # (own process: it patches the engine factory)
    mp.spawn(_driver_rank, args=(world, _free_port(), root, ck, ttype, out), nprocs=world, join=True)
    one = torch.load(os.path.join(out, "w1_rank0.pt"), weights_only=False)
    rs = [torch.load(os.path.join(out, "w%d_rank%d.pt" % (world, r)), weights_only=False) for r in range(world)]
    r0, r1 = rs[0], rs[1]
    for rr in rs[1:]:
        assert torch.equal(r0["wi"], rr["wi"])
        for k in r0["theta"]:
            assert torch.equal(r0["theta"][k], rr["theta"][k]), k
    assert all(rr["wu"].shape[0] == U // world for rr in rs)                # each rank holds its own user rows only
    num = re.compile(r"-?\d+\.\d+(?:e-?\d+)?")
    strip = lambda t: [re.sub(r"\s+", " ", l).replace("[ ", "[").replace(" ]", "]") for l in t.splitlines() if "time cost" not in l]
    a, b = strip(one["log"]), strip(r0["log"])
    assert [num.sub("#", l) for l in a] == [num.sub("#", l) for l in b]
    for la, lb in zip(a, b):
        for x, y in zip(num.findall(la), num.findall(lb)):
            x, y = float(x), float(y)
            tol = 2.0 / 160 + 1e-4 if ("recall" in la or "reacll" in la or "ndcg" in la) else 1e-4 * max(1.0, abs(x))
            assert abs(x - y) <= tol, (la, lb)
    np.testing.assert_allclose(torch.cat([rr["wu"] for rr in rs]).numpy(), one["wu"].numpy(), rtol=5e-3, atol=5e-4)
    np.testing.assert_allclose(r0["wi"].numpy(), one["wi"].numpy(), rtol=5e-3, atol=5e-4)
    for rr in rs[1:]:
        assert strip(rr["log"]) == strip(r0["log"])  # every rank holds the job's numbers (cli.main silences all but rank 0)


def test_item_shard_layout_routes_every_row_to_exactly_one_place():
    """Routing of the item-sharded bare step (sml_amd.dist.item_shard_layout / item_owner; the device code applies the
    same arithmetic): the head is replicated (owner -1), every tail row has exactly one owner and a local row inside that
    owner's shard, the shards cover the table, and the layout degenerates properly (no head; all head; one rank)."""
    from sml_amd import dist as SD
    for n_item, world, head in ((120, 2, 16), (1000003, 8, 4096), (77, 8, 0), (50, 4, 50), (50, 4, 999), (9, 1, 3)):
        H, S = SD.item_shard_layout(n_item, world, head)
        assert 0 <= H <= n_item and S >= 1 and H + world * S >= n_item and H + world * S - n_item < world + S
        rows = np.arange(n_item, dtype=np.int64)
        owner, local = SD.item_owner(rows, world, H, S)
        assert np.all(owner[:H] == -1) and np.array_equal(local[:H], rows[:H])
        assert np.all((owner[H:] >= 0) & (owner[H:] < world)) and np.all((local[H:] >= 0) & (local[H:] < S))
        back = np.where(owner < 0, local, H + owner * S + local)
        assert np.array_equal(back, rows)                                 # a bijection onto (owner, local)
        counts = np.bincount(owner[H:], minlength=world) if n_item > H else np.zeros(world, dtype=np.int64)
        assert counts.max() <= S and (counts > 0).sum() == min(world, -(-(n_item - H) // S) if n_item > H else 0)


# ----------------------------------------------------------------------------- the rank launcher (bench.py / main_yelp.py --gpus N)
def test_launcher_starts_fresh_ranks_with_the_ipc_environment_and_relays_rank_zero():
    """sml_amd.launch.spawn_ranks: N fresh rank processes from a parent that makes no GPU call, the rendezvous and
    HSA_ENABLE_IPC_MODE_LEGACY=0 in their environment, rank 0's stdout is the job's, the others' goes to stderr."""
    import json
    from sml_amd import launch
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_child.py")
    code, out = launch.spawn_ranks([sys.executable, child], 2, echo_stdout=False, timeout=180)
    assert code == 0
    # (gloo itself prints a connection banner on stdout; bench.py keeps such library chatter off the job's stdout with
    # its fd-level redirect -- this child does not, so the JSON line is picked out)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "noise from rank" not in out, out
    got = json.loads(lines[0])
    assert got == {"sum": 3.0, "world": 2, "ipc_legacy": "0", "launched": "1", "local_rank": "0", "one_device": None, "master": "127.0.0.1"}
    code, out = launch.spawn_ranks([sys.executable, child], 2, one_device=True, echo_stdout=False, timeout=180)
    assert code == 0 and json.loads([l for l in out.splitlines() if l.startswith("{")][0])["one_device"] == "1"


def test_launcher_returns_the_failing_ranks_code_and_stops_the_others():
    from sml_amd import launch
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_child.py")
    code, out = launch.spawn_ranks([sys.executable, child, "--fail-rank", "1"], 2, echo_stdout=False, timeout=180)
    assert code == 3
    assert "{" not in out             # rank 0 never got past the barrier: no line, and the launcher did not hang


def test_bench_parent_makes_no_gpu_call_before_spawning(monkeypatch):
    """`python bench.py --gpus 2` in a process that is not a rank must hand over to the launcher before anything touches
    torch.cuda (the parent of GPU ranks may not initialise HIP: an exec / fork after that takes the box down)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from sml_amd import launch
    seen = {}

    def fake_spawn(argv, world, one_device=False, **kw):
        seen.update(argv=argv, world=world, one_device=one_device)
        return 0, ""
    monkeypatch.setattr(launch, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU call in the parent")))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU call in the parent")))
    # (torch.cuda.device_count() falls back to hipGetDeviceCount when amdsmi is unavailable -- the parent counts
    # GPUs through sysfs instead)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("GPU call in the parent")))
    if hasattr(torch._C, "_cuda_getDeviceCount"):
        monkeypatch.setattr(torch._C, "_cuda_getDeviceCount", lambda: (_ for _ in ()).throw(AssertionError("GPU call in the parent")))
    for k in ("SML_LAUNCHED", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--one-device"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert seen["world"] == 2 and seen["one_device"] is True and seen["argv"][2:] == ["--gpus", "2", "--steps", "1", "--one-device"]


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    with open("/proc/%d/stat" % pid) as f:           # (a zombie is not alive)
        return f.read().rsplit(")", 1)[1].split()[0] != "Z"


def _wait_pids(base, world, timeout=60.0):
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        if all(os.path.exists("%s.%d" % (base, r)) and os.path.getsize("%s.%d" % (base, r)) > 0 for r in range(world)):
            return [int(open("%s.%d" % (base, r)).read()) for r in range(world)]
        time.sleep(0.1)
    raise AssertionError("ranks did not start")


def test_launcher_time_out_ends_a_hung_job_and_leaves_no_rank_behind(tmp_path):
    """a rank that hangs must not hold the job forever -- the time-out stops every rank (exit code 124)."""
    import threading
    from sml_amd import launch
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_child.py")
    base = str(tmp_path / "pid")
    res = {}
    th = threading.Thread(target=lambda: res.update(r=launch.spawn_ranks([sys.executable, child, "--hang", base], 2, echo_stdout=False, timeout=20)))
    th.start()
    pids = _wait_pids(base, 2)
    th.join(timeout=90)
    assert not th.is_alive()
    code, out = res["r"]
    assert code == 124 and "first line of a job that hangs" in out
    assert not any(_alive(p) for p in pids)


@pytest.mark.parametrize("sig", ["TERM", "KILL"])
def test_launcher_that_is_signalled_takes_its_ranks_with_it(tmp_path, sig):
    """SIGTERM to the launcher (a scheduler, a test harness's time-out) -> its handler stops the ranks;
    SIGKILL -> PR_SET_PDEATHSIG does.  Rank 0's output has been relayed line by line before that (not buffered to the end)."""
    import signal
    import subprocess
    import time
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_child.py")
    base = str(tmp_path / "pid")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys; sys.path.insert(0, %r); from sml_amd import launch; "
            "code, _ = launch.spawn_ranks([sys.executable, %r, '--hang', %r], 2); sys.exit(code)" % (repo, child, base))
    p = subprocess.Popen([sys.executable, "-c", prog], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        pids = _wait_pids(base, 2)
        assert p.stdout.readline().strip() == "first line of a job that hangs"      # streamed while the job still runs
        assert all(_alive(q) for q in pids)
        p.send_signal(signal.SIGTERM if sig == "TERM" else signal.SIGKILL)
        rc = p.wait(timeout=60)
        assert rc == (128 + signal.SIGTERM if sig == "TERM" else -signal.SIGKILL)
        t0 = time.time()
        while any(_alive(q) for q in pids) and time.time() - t0 < 30:
            time.sleep(0.1)
        assert not any(_alive(q) for q in pids)
    finally:
        if p.poll() is None:
            p.kill()
        for r in range(2):
            f = "%s.%d" % (base, r)
            if os.path.exists(f):
                try:
                    os.kill(int(open(f).read()), signal.SIGKILL)
                except (ProcessLookupError, ValueError):
                    pass


def test_visible_gpus_reads_sysfs_and_never_calls_hip(monkeypatch):
    from sml_amd import launch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("GPU call")))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    n = launch.visible_gpus()
    assert n is None or n >= 0
    if n is not None:               # every variable torch on ROCm honours narrows the count; set-and-empty hides all devices
        monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "0")
        assert launch.visible_gpus() == min(n, 1)
        monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
        assert launch.visible_gpus() == 0
        monkeypatch.delenv("CUDA_VISIBLE_DEVICES")
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
        assert launch.visible_gpus() == min(n, 3)
    assert launch.job_timeout(None, 3600.0) == 3600.0 and launch.job_timeout(0, 5.0) is None and launch.job_timeout("7") == 7.0

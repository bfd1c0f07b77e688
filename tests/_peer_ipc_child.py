"""One rank of a TWO-PROCESS job sharing one GPU (test infrastructure, started by tests/test_hip_parity.py): the one-shot peer
exchange through REAL hipIpc mappings -- sml_peer_export / sml_peer_open of the uncached inbox and flags regions of
another process -- with gloo carrying the set-up (handles, barriers).  Each process has its own HIP context and queues, as
in the one-process-per-GPU deployment; what differs from an 8-GPU node is only that both inboxes live on the same device.

    python tests/_peer_ipc_child.py RANK WORLD PORT OUT_DIR
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def workload():
    """The inputs of test_two_ranks_on_one_gpu_mf_and_tr_stage_equal_the_single_engine_run (same seeds)."""
    torch.manual_seed(5)
    U, I, d, B, n = 200, 120, 32, 64, 300
    wu, wi = torch.randn(U, d) * 0.3, torch.randn(I, d) * 0.3
    u = torch.randint(0, U, (n,)); u[:9] = 3
    u[2 * B:3 * B] = torch.randint(0, 100, (B,))            # batch 2: every user belongs to rank 0 -> rank 1's batch is empty
    tri = torch.stack([u, torch.randint(0, I, (n,)), torch.randint(0, I, (n,))], 1)
    tri[5, 2] = tri[5, 1]
    return U, I, d, B, n, wu, wi, tri


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    from conftest import make_mf, make_transfer
    from sml_amd import dist as SD
    from sml_amd.engine import HipEngine
    dev = "cuda:0"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    U, I, d, B, n, wu, wi, tri = workload()
    sd = torch.load(os.path.join(out_dir, "theta0.pt"))
    e = HipEngine(dev, d, B)
    ok = SD.peer_setup(e, dist, None, rows_cap=2 * B, timeout_s=20.0)      # hipIpc handles travel over gloo
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok))
    assert all(oks), "peer self-check failed: %s" % oks
    ctx = SD.DistContext(dist, e.device)
    ctx.mode, ctx.native, ctx.peer_rows_cap = "peer", True, 2 * B
    e.dist, e.grad_hook = ctx, None
    # a large all-reduce through the mapped inboxes first (the whole theta slot)
    big = torch.arange(8192, device=dev, dtype=torch.float32) * (rank + 1)
    got = e.peer_allreduce_check(big, timeout_s=20.0)
    assert torch.equal(got, torch.arange(8192, device=dev, dtype=torch.float32) * (world * (world + 1) // 2))
    lo, hi_ = SD.user_range(U, world, rank)
    lu, li = wu * 0.9, wi * 0.9
    m = make_mf(hi_ - lo, I, d, wu[lo:hi_].numpy(), wi.numpy(), device=dev)
    net = make_transfer(d, device=dev)
    net.load_state_dict(sd)
    route = ctx.route_epoch(tri.numpy(), B, U, mean_loss=True)
    a = e.mf_stage_epoch(m, net, lu[lo:hi_].to(dev), li.to(dev), route.local_tri, route.cap, 0.01, 1e-6,
                         plan=route.plan, exchange=route.exchange(d))
    e.mf_flush(m)
    hu_, hi2 = m.user_laten.weight.detach().clone(), m.item_laten.weight.detach().clone()
    b = e.tr_stage_epoch(net, lu[lo:hi_].to(dev), li.to(dev), hu_, hi2, route.local_tri, route.cap, 1e-3, 1e-4, plan=route.plan)
    torch.cuda.synchronize()
    timeouts = e.peer_status()
    dist.barrier()                    # nobody unmaps an inbox a peer may still be writing into
    torch.save(dict(l_mf=a.cpu().numpy(), l_tr=b.cpu().numpy(), wu=hu_.cpu(), wi=hi2.cpu(), timeouts=timeouts,
                    theta={k: v.detach().cpu().clone() for k, v in net.state_dict().items()}),
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

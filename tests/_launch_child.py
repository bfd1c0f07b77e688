"""One rank of a job started by sml_amd.launch.spawn_ranks (test infrastructure): joins a gloo group from the environment
the launcher made, all-reduces, and rank 0 prints ONE JSON line -- the shape of bench.py under `--gpus N`, without a GPU.

    python tests/_launch_child.py [--fail-rank R] [--hang PIDFILE]

--hang: every rank writes its PID to PIDFILE.<rank>, rank 0 prints a first line, and all sleep "forever" (a hung job: the
launcher's time-out / signal handling has to end them).
"""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fail = int(sys.argv[sys.argv.index("--fail-rank") + 1]) if "--fail-rank" in sys.argv else -1
    if "--hang" in sys.argv:
        import time
        with open("%s.%d" % (sys.argv[sys.argv.index("--hang") + 1], rank), "w") as f:
            f.write(str(os.getpid()))
        if rank == 0:
            print("first line of a job that hangs", flush=True)
        time.sleep(600)
        return
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == fail:
        print("rank %d fails on purpose" % rank, file=sys.stderr)
        os._exit(3)
    if rank != 0:
        print("noise from rank %d" % rank)          # must not reach the job's stdout
    dist.barrier()
    if rank == 0:
        print(json.dumps({"sum": float(t.item()), "world": world, "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "launched": os.environ.get("SML_LAUNCHED"), "local_rank": os.environ.get("LOCAL_RANK"),
                          "one_device": os.environ.get("SML_ONE_DEVICE"), "master": os.environ.get("MASTER_ADDR")}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

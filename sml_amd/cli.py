"""Command line of main_yelp.py / main_news.py (reference main_yelp.py:10-172,
main_news.py:8-232): same flags, defaults, type quirks, seeding order and banner."""
import argparse
import os

import numpy as np
import torch

# (flag, kwargs) in the reference's order.  Quirks kept on purpose: the type=bool flags are
# True for ANY non-empty string, and Load_W_hat / clip_grad / need_adaptive have no type at
# all (a value given on the command line arrives as a non-empty, hence truthy, string).
_COMMON = [
    ("--data_path", dict(default='/home/sml/dataset/', help='dataset path')),
    ("--multi_num", dict(type=int, default=10, help='outer loop count (stop condition of SML)')),
    ("--MF_lr", dict(type=float, default=0.01, help='learning rate of the MF step')),
    ("--MF_epochs", dict(type=int, default=1, help='epochs of the MF step')),
    ("--l2", dict(type=float, default=1e-6, help='L2 weight of the MF step')),
    ("--MF_batch_size", dict(type=int, default=1024, help='batch size of the MF step')),
    ("--laten", dict(type=int, default=64, help='embedding width')),
    ("--pre_model", dict(default=None, help='pretrained MF model (whole-module pickle)')),
    ("--MF_sample", dict(default="all", help='MF negative sampling: all or alone')),
    ("--Load_W_hat", dict(default=False, help='reload W_hat into MF after transfer training')),
    ("--clip_grad", dict(default=False, help='(unused in the final reference version)')),
    ("--need_adaptive", dict(default=False, help='(unused in the final reference version)')),
    ("--maxnorm_grad", dict(type=float, default=3.0, help='(unused in the final reference version)')),
    ("--TR_lr", dict(type=float, default=0.001, help='learning rate of the transfer step')),
    ("--TR_l2", dict(type=float, default=0.0001, help='weight decay of the transfer step')),
    ("--TR_epochs", dict(type=int, default=1, help='epochs of the transfer step')),
    ("--TR_batch_size", dict(type=int, default=256, help='batch size of the transfer step')),
    ("--TR_sample_type", dict(default="alone", help='negatives of the transfer step: all or alone')),
    ("--TR_with_MF_bias", dict(type=bool, default=False, help='feed the MF bias to the transfer net')),
    ("--TR_stop_", dict(type=bool, default=False, help='freeze the transfer net during the test periods')),
    ("--transfer_type", dict(default="conv_com", help='transfer architecture: conv_com (the paper\'s) or conv')),
    ("--seed", dict(type=int, default=2000, help='random seed')),
    ("--numworkers", dict(type=int, default=4, help='accepted for compatibility; batches are built in-process')),
    ("--cuda", dict(type=int, default=0, help='which GPU')),
    ("--topK", dict(type=int, default=20, help='K of the in-training recall/ndcg prints')),
    ("--pass_num", dict(type=int, default=1, help='offline passes (1 in the final reference version)')),
    ("--norm", dict(type=bool, default=False, help='(unused in the final reference version)')),
    ("--Lambda_lr", dict(type=float, default=0.01, help='(unused)')),
    ("--min_l2", dict(type=float, default=0.0001, help='(unused)')),
    ("--set_t_as_tt", dict(type=bool, default=False, help='(unused)')),
    ("--tqdm", dict(type=bool, default=False, help='(unused)')),
    ("--need_writer", dict(type=bool, default=False, help='tensorboard summaries')),
    ("--test_in_TR_Train", dict(type=bool, default=False, help='(unused)')),
    # extension (not a reference flag): draw the transfer stage's shuffles and negatives on the GPU -- the same
    # distribution, not the reference's numpy/torch random streams (default: stream-exact host path)
    ("--device_batches", dict(type=int, default=0, help='1: build the MF and transfer stage batches on the device (not stream-exact)')),
    # extension (the reference is single-device, main_yelp.py:125): run as N rank processes, one per GPU of this node.
    # The process that is given --gpus N > 1 starts the ranks itself (sml_amd.launch) and only relays rank 0's output
    ("--gpus", dict(type=int, default=1, help='N > 1: one rank process per GPU of this node (users row-sharded by owner)')),
    ("--job_timeout", dict(type=float, default=None, help='--gpus N > 1: seconds before the launcher stops the ranks and exits 124 '
                                                          '(default: SML_JOB_TIMEOUT_S, else no limit)')),
]

_PER_DATASET = {
    "yelp": dict(data_name="yelp", multi_num=10, MF_epochs=1, TR_epochs=1,
                 pre_model="/home/sml/save_model/sml/yelp/BCE_init.pkl",
                 periods=40, train_from=10, test_from=30),
    "news": dict(data_name="news", multi_num=7, MF_epochs=2, TR_epochs=2,
                 pre_model="/home/sml/save_model/sml/news/BCE_init.pkl",
                 periods=63, train_from=21, test_from=48),
}


def get_parse(which="yelp"):
    cfg = _PER_DATASET[which]
    parser = argparse.ArgumentParser(description='MF and TR(transfer) parameters in our SML.')
    parser.add_argument('--data_name', default=cfg["data_name"], help='dataset name: yelp or news (i.e Adressa)')
    for flag, kw in _COMMON:
        kw = dict(kw)
        name = flag[2:]
        if name in cfg:
            kw["default"] = cfg[name]
        parser.add_argument(flag, **kw)
    return parser


def _banner(args, which):
    stars = "*********************  parameters information ****************************************" \
        if which == "yelp" else "**********  SML parameters ***************"
    print(stars)
    print(args)
    print("stop:", args.TR_stop_)
    if args.Load_W_hat:
        print("load w_hat:", args.Load_W_hat)
    print("TR sample type", args.TR_sample_type)
    print("MF sample type:", args.MF_sample)
    print("MF: lr:{},l2:{},batch_size:{},laten:{},epoch:{}".format(args.MF_lr, args.l2, args.MF_batch_size,
                                                                     args.laten, args.MF_epochs))
    print("TR: lr:{},l2:{},batch_size:{},epoch:{}".format(args.TR_lr, args.TR_l2, args.TR_batch_size, args.TR_epochs))
    print("top k:", args.topK)
    print(stars + ("\n" if which == "yelp" else ""))


def main(which, argv=None):
    if os.environ.get("SML_FAULT_DUMP_S"):          # debugging aid: every thread's Python stack to stderr after that many seconds (a hung job says where)
        import faulthandler
        import sys
        faulthandler.dump_traceback_later(float(os.environ["SML_FAULT_DUMP_S"]), repeat=False, file=sys.stderr)
    from data import dataset2
    from model import transfer

    cfg = _PER_DATASET[which]
    args = get_parse(which).parse_args(argv)
    from . import launch
    if args.gpus > 1 and not launch.is_rank_process():
        # the parent of the job: NOTHING here has touched the GPU yet.  Start one fresh rank process per GPU (IPC mode and
        # rendezvous in their environment) running this same command line, relay rank 0's output, leave with their code
        import sys
        script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "main_%s.py" % which)
        rest = list(sys.argv[1:] if argv is None else argv)
        code, _ = launch.spawn_ranks([sys.executable, script] + rest, args.gpus, one_device=os.environ.get("SML_ONE_DEVICE") == "1",
                                      timeout=launch.job_timeout(args.job_timeout))
        raise SystemExit(code)
    if which == "yelp" and "LOCAL_RANK" not in os.environ:
        os.environ["CUDA_VISIBLE_DEVICES"] = str(args.cuda)      # reference main_yelp.py:125
    # One process per GPU (`main_yelp.py --gpus N`, or `torchrun --nproc-per-node N main_yelp.py ...`).  Every rank runs
    # this same program on the same seeds and files; users are sharded by owner inside meta_train (sml_amd/dist.py), rank 0 prints.
    dist = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        launch.prepare_rank_env()            # (torchrun does not set the IPC mode: before the first HIP call of this process)
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(launch.backend() if torch.cuda.is_available() else "gloo")
        if dist.get_rank() != 0:
            import sys
            sys.stdout = open(os.devnull, "w")
    # The host side of this process only draws random numbers and slices index arrays; torch's CPU
    # thread pool costs tens of ms per randperm(n > 32768) to wake (it splits the arange fill), so keep
    # host torch ops on the calling thread.  Results do not depend on the thread count.
    torch.set_num_threads(int(os.environ.get("SML_HOST_THREADS", "1")))
    if which == "yelp":
        print("###(multi num,l2)", 10, 1e-06)                     # the reference's (unused) sweep banner
    # seeding order of the reference (main_yelp.py:137-139)
    torch.manual_seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(args.seed + 1)
    np.random.seed(args.seed + 2)
    _banner(args, which)
    file_list = [str(i) for i in range(0, cfg["periods"])]
    test_list = [str(j) for j in range(cfg["test_from"], cfg["periods"])]
    sets = dataset2.transfer_data(args, path=args.data_path, datasetname=args.data_name, file_path_list=file_list,
                                  test_list=test_list, validation_list=None,
                                  online_train_time=round(cfg["train_from"]), online_test_time=round(cfg["test_from"]))
    meta = transfer.meta_train(args, sets, sets.user_number, sets.item_number, args.laten) if dist is None else \
        transfer.meta_train(args, sets, sets.user_number, sets.item_number, args.laten, dist=dist)
    # (what exists now -- torch, the model, the period loader -- is set aside from the garbage collector's full collections: one of
    # them in the middle of a stage is 40-60 ms during which the host queues nothing)
    import gc
    gc.collect()
    gc.freeze()
    meta.run(args)
    if which == "yelp":
        print("@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@@TR:L2:", 1e-06)
    print("##")
    print("##")
    print("\n")            # both mains end this way (main_yelp.py:172, main_news.py:232)
    return meta

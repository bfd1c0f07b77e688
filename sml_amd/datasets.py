"""Period data source and batch supply (reference data/dataset2.py, data/dataset.py).

The classes keep the reference's names and per-item semantics, and add a
vectorised `epoch_triples(order)` that yields a whole epoch's (user, item, neg)
triples at once with the SAME random-number consumption as a
DataLoader(shuffle=True, num_workers=0) pass over `__getitem__`, so the host
never loops over single items in Python.  `loader_order(n)` reproduces the two
draws such a DataLoader pass makes from torch's global generator.
"""
import copy
import os

import numpy as np
import torch
from torch.utils.data import Dataset


_TORCH_DRAWS = [0]          # draws this module made from torch's global generator (EpochSpeculation predicts the next pass's shuffle from it)


def torch_draws():
    return _TORCH_DRAWS[0]


def loader_base_seed_draw():
    """The draw every DataLoader iteration makes on creation (_BaseDataLoaderIter.__init__)."""
    _TORCH_DRAWS[0] += 1
    return int(torch.empty((), dtype=torch.int64).random_().item())


def loader_order(n, shuffle=True):
    """Index order of one DataLoader(num_workers=0) pass over n items, consuming torch's
    global RNG exactly as DataLoader + RandomSampler do (base seed, then sampler seed)."""
    loader_base_seed_draw()
    if not shuffle:
        return np.arange(n, dtype=np.int64)
    seed = loader_base_seed_draw()          # (the sampler's seed: the same kind of draw)
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).numpy()


def _gather_into(tri, ui, order, mat=None, col=0):
    """tri[:, :2] = ui[order] (and tri[:, 2] = mat[order, col]) through the compiled loops of libsml_hip.so
    (sml_host_gather_pairs / sml_host_gather_column): numpy's fancy indexing builds a temporary and copies it through a
    strided view -- 1.9 ms of a 75,000-row epoch's 4.4 ms on the host.  False when the arrays are not in a layout the
    loops take (the caller then indexes with numpy)."""
    from . import _lib
    order = np.asarray(order)
    if (order.dtype != np.int64 or not order.flags.c_contiguous or ui.dtype != np.int64 or not ui.flags.c_contiguous
            or tri.dtype != np.int64 or not tri.flags.c_contiguous):
        return False
    if mat is not None and (mat.ndim != 2 or mat.dtype.kind != "i" or mat.dtype.itemsize not in (4, 8) or mat.strides[1] != mat.dtype.itemsize):
        return False
    lib = _lib.load()
    n = order.shape[0]
    _lib.check(lib.sml_host_gather_pairs(ui.ctypes.data, ui.shape[0], order.ctypes.data, n, tri.ctypes.data), "sml_host_gather_pairs")
    if mat is not None:
        _lib.check(lib.sml_host_gather_column(mat.ctypes.data, mat.shape[0], mat.strides[0], mat.dtype.itemsize, col, order.ctypes.data,
                                              n, tri.ctypes.data, 2), "sml_host_gather_column")
    return True


class testDataset(Dataset):
    """reference data/dataset2.py:160-170"""

    def __init__(self, dataset):
        super(testDataset, self).__init__()
        self.data = dataset

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        return self.data[idx]


class trainDataset_withPreSample(Dataset):
    """Rows [user, item, c2, c3, ...] with pre-sampled negatives; each full pass over the
    data uses ONE column as the negative (reference data/dataset2.py:172-201).  Quirk kept:
    the candidate columns are 1..C-1, so column 1 (the positive itself) can be drawn."""

    def __init__(self, input_dataset):
        super(trainDataset_withPreSample, self).__init__()
        self.all_data = input_dataset      # read-only here (the reference deep-copies 0.6 GB per phase for nothing)
        self.data_len = input_dataset.shape[0]
        self.neg_all = input_dataset.shape[1] - 2
        self.neg_flag = np.arange(1, self.all_data.shape[1])
        np.random.shuffle(self.neg_flag)
        self.used_neg_count = 0
        self.have_read = 0
        self._ui = None                    # contiguous int64 copy of the (user, item) columns, made on the first whole pass

    def __len__(self):
        return self.data_len

    def _advance(self, reads):
        self.have_read += reads
        if self.have_read >= self.data_len:
            self.have_read = 0
            self.used_neg_count += 1
            if self.used_neg_count >= self.neg_all:
                np.random.shuffle(self.neg_flag)
                self.used_neg_count = 0

    def __getitem__(self, idx):
        row = self.all_data[idx]
        out = (row[0], row[1], row[self.neg_flag[self.used_neg_count]])
        self._advance(1)
        return out

    def epoch_triples(self, order):
        """One full pass in `order` -> int64 [n,3]."""
        if self.have_read != 0 or len(order) != self.data_len:
            raise ValueError("epoch_triples needs a whole pass starting at a pass boundary")
        col = self.neg_flag[self.used_neg_count]
        a = self.all_data                    # gather columns, not 1001-wide rows
        if self._ui is None:                 # (user, item) side by side: one 16-byte gather per row and epoch instead of two
            self._ui = np.ascontiguousarray(a[:, :2], dtype=np.int64)      # strided ones through the 8 KB rows
        tri = np.empty((self.data_len, 3), dtype=np.int64)
        if not _gather_into(tri, self._ui, order, a, int(col)):
            tri[:, :2] = self._ui[order]
            tri[:, 2] = a[order, col]
        self._advance(self.data_len)
        return tri

    def epoch_triples_device(self, engine, seed):
        """One full pass assembled ON THE DEVICE (fast mode, --device_batches): the pass's negative column -- chosen exactly as
        epoch_triples chooses it: neg_flag is this object's numpy shuffle -- crosses PCIe once as a contiguous vector
        (n * 8 bytes; the rows themselves never do), the shuffle is the counter-based permutation of sml_device_epoch
        (the same distribution as DataLoader(shuffle=True), not torch's randperm stream), the (user, item) columns are
        resident.  Returns an int64 [n,3] device tensor.  Every rank of a job derives the same epoch from the same seed."""
        if self.have_read != 0:
            raise ValueError("epoch_triples_device needs a whole pass starting at a pass boundary")
        dev = engine.device
        col = int(self.neg_flag[self.used_neg_count])
        a = self.all_data
        if self._ui is None:
            self._ui = np.ascontiguousarray(a[:, :2], dtype=np.int64)
        cache = self.__dict__.setdefault("_dev_ui", {})
        ui = cache.get(str(dev))
        if ui is None or ui[0] is not self._ui:
            ui = cache[str(dev)] = (self._ui, torch.from_numpy(self._ui).pin_memory().to(dev, non_blocking=True))
        colv = torch.from_numpy(np.ascontiguousarray(a[:, col], dtype=np.int64)).pin_memory().to(dev, non_blocking=True)
        out = engine.device_epoch(ui[1], self.data_len, seed, mat=colv, row_stride=1, col=0)
        self._advance(self.data_len)
        return out


class offlineDataset_withsample(Dataset):
    """(user, item) pairs; the negative is drawn per access, uniformly from the items that
    occur in this set, rejecting the user's own items (reference data/dataset.py:41-71)."""

    def __init__(self, dataset):
        super(offlineDataset_withsample, self).__init__()
        self.user = dataset[:, 0]
        self.item = dataset[:, 1]
        print("user max:", self.user.max())
        print("user max:", self.item.max())     # (sic) the reference prints the item max under this label
        self.item_all = np.unique(self.item)
        self._user_list = None                  # per-item access only: built on first use
        self._stride = int(self.item_all.max()) + 1
        self._pairs = np.unique(self.user.astype(np.int64) * self._stride + self.item.astype(np.int64))
        # the same pairs in CSR form (user -> its items, ascending) for the compiled sequential resolver
        pu = self._pairs // self._stride
        self._n_users = int(pu.max()) + 1 if pu.size else 0
        self._uptr = np.ascontiguousarray(np.searchsorted(pu, np.arange(self._n_users + 1)), dtype=np.int64)
        self._uitems = np.ascontiguousarray(self._pairs % self._stride, dtype=np.int64)
        self._ui = np.ascontiguousarray(np.stack([self.user, self.item], axis=1), dtype=np.int64)    # epoch_triples gathers both at once

    @property
    def user_list(self):
        if self._user_list is None:
            ul = {}
            for u, i in zip(self.user.tolist(), self.item.tolist()):
                ul.setdefault(u, []).append(i)
            self._user_list = ul
        return self._user_list

    def __len__(self):
        return self.user.shape[0]

    def __getitem__(self, idx):
        user, item = self.user[idx], self.item[idx]
        neg = np.random.choice(self.item_all, 1)[0]
        while neg in self.user_list[user]:
            neg = np.random.choice(self.item_all, 1)[0]
        return (user, item, neg)

    def epoch_triples(self, order, rng=None):
        """(rng: a numpy RandomState to draw from instead of the global one -- EpochSpeculation's private copy.)
        Same triples, and same numpy global-RNG end state, as calling __getitem__ for every
        index of `order` in turn.  np.random.choice(a, 1) is one legacy randint(0, len(a)) draw and a
        block of such draws is the same stream, so candidates are drawn in blocks and the sequential
        accept/reject walk over them runs in compiled code (sml_host_resolve_negatives_csr, a host-side
        helper of libsml_hip.so).  Each block holds exactly one candidate per still-unresolved element
        -- every one of which the per-item loop would draw too -- so the generator ends where it would."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        order = np.asarray(order)
        n = order.shape[0]
        out = np.empty((n, 3), dtype=np.int64)
        if not _gather_into(out, self._ui, order):
            out[:, :2] = self._ui[order]
        users = np.ascontiguousarray(out[:, 0])
        pop = self.item_all.shape[0]
        items_all = np.ascontiguousarray(self.item_all, dtype=np.int64)
        negs = np.empty(n, dtype=np.int64)
        used, got = ctypes.c_int64(0), ctypes.c_int64(0)
        done, drawn = 0, 0
        while done < n:
            k = n - done
            cand = np.ascontiguousarray(items_all[(rng if rng is not None else np.random).randint(0, pop, size=k)])
            rc = lib.sml_host_resolve_negatives_csr(users.ctypes.data + 8 * done, k, cand.ctypes.data, k,
                                                    self._uptr.ctypes.data, self._n_users, self._uitems.ctypes.data,
                                                    negs.ctypes.data + 8 * done, ctypes.byref(used), ctypes.byref(got))
            _lib.check(rc, "sml_host_resolve_negatives_csr")
            done += got.value
            drawn += k
            if drawn > 64 * (n + 64):
                raise RuntimeError("negative sampling does not terminate: a user owns (almost) every item")
        out[:, 2] = negs
        return out


    def epoch_triples_device(self, engine, seed):
        """One shuffled pass with fresh negatives, built ON THE DEVICE (fast mode: the same distribution as
        epoch_triples, not the reference's random streams): the counter-based permutation of sml_device_epoch over the
        resident (user, item) pairs, then engine.sample_negatives.  Returns an int64 [n,3] device tensor; nothing crosses
        PCIe per epoch, and every rank of a job derives the same epoch from the same seed."""
        import torch
        dev = engine.device
        cache = self.__dict__.setdefault("_dev_cache", {})
        c = cache.get(str(dev))
        if c is None:
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
            c = cache[str(dev)] = dict(user=t(self.user), item=t(self.item), item_all=t(self.item_all), uptr=t(self._uptr),
                                       uitems=t(self._uitems), gen=torch.Generator(device=dev))
        if "ui" not in c:
            c["ui"] = torch.stack([c["user"], c["item"]], dim=1).contiguous()
        # the shuffle: sml_device_epoch's counter-based permutation (columns 0, 1 of the triples); then the negatives
        tri = engine.device_epoch(c["ui"], len(self), seed)
        users = tri[:, 0].contiguous()
        negs, failed = engine.sample_negatives(users, c["item_all"], c["uptr"], c["uitems"], int(seed) + 1)
        self._last_failed = failed            # device counter; checked lazily by the caller if it cares
        tri[:, 2] = negs
        return tri


def _same_rng_state(a, b):
    return (a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and np.array_equal(a[1], b[1]))


class EpochSpeculation(object):
    """The NEXT pass of an offlineDataset_withsample, drawn ahead on a helper thread while the host queues the current one.

    The reference-exact batch supply costs the host 3.5 ms per 75,000-triple pass (the sequential accept / reject walk of the
    negatives), and the real driver is host-bound by about that much per phase (tools/time_driver_stage.py --profile).  The
    pass is a pure function of (torch's shuffle, numpy's generator state at the pass's first draw), so it can be computed
    early from PREDICTED inputs and adopted only if the prediction was right:
      * torch: the global generator is saved, the draws the driver is expected to make before the pass (`torch_draws_before`:
        DataLoader base seeds and sampler seeds of the MF epochs and of every test in between -- the driver counts them from
        one phase to the next) are made, the pass's own order is drawn, the generator is restored;
      * numpy: a private RandomState starts from the global state, makes the draws expected before the pass (the
        pre-sampled MF dataset's constructor shuffles its column order), and the helper thread samples from it;
      * adopt(): taken only if the order the driver really drew equals the predicted one AND the global numpy state equals
        the state the helper started from; the global state is then set to where the helper ended.  Anything else (another
        consumer of either generator in between, another dataset) discards the work and the caller samples in place.
    Either way triples, logs and generator states are exactly those of the in-place path (tests/test_host_logic.py)."""

    def __init__(self, train_set, torch_draws_before=0, np_shuffles_before=()):
        import threading
        self.train_set = train_set
        t_state, counted = torch.get_rng_state(), _TORCH_DRAWS[0]
        try:
            for _ in range(int(torch_draws_before)):
                loader_base_seed_draw()
            self.order = loader_order(len(train_set), shuffle=True)
        finally:
            torch.set_rng_state(t_state)
            _TORCH_DRAWS[0] = counted
        self.rs = np.random.RandomState()
        self.rs.set_state(np.random.get_state())
        for k in np_shuffles_before:
            self.rs.shuffle(np.arange(1, int(k) + 1))
        self.state_before = self.rs.get_state()
        self.triples, self.error = None, None
        self.thread = threading.Thread(target=self._run, name="sml-epoch-speculation", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            self.triples = self.train_set.epoch_triples(self.order, rng=self.rs)
        except BaseException as e:          # surfaces as "not adopted": the caller samples in place and raises there
            self.error = e

    def adopt(self, train_set, order):
        """The pass's triples if the prediction held (global numpy state advanced as the in-place path would), else None."""
        self.thread.join()
        if (self.error is not None or self.triples is None or train_set is not self.train_set
                or not np.array_equal(np.asarray(order), self.order)
                or not _same_rng_state(np.random.get_state(), self.state_before)):
            return None
        np.random.set_state(self.rs.get_state())
        return self.triples


class transfer_data(object):
    """Per-stage period data (reference data/dataset2.py:203-351).

    Layout under path/datasetname/: information.npy = [n_interactions, n_user, n_item],
    train/<p>.npy int [n,2], test/<p>.npy int [n, 2+neg]."""

    def __init__(self, args, path="dataset/", datasetname="News", online_train_time=21, file_path_list=None,
                 test_list=None, validation_list=None, online_test_time=48):
        self.TR_sample_type = args.TR_sample_type
        self.TR_stop_ = args.TR_stop_
        self.MF_sample = args.MF_sample
        self.current_as_set_tt = args.set_t_as_tt
        self.path = path
        self.dataname = datasetname
        self.file_list = file_path_list
        self.test_list = test_list
        self.val_list = validation_list
        self.len = len(file_path_list)
        self.online_trian_time = online_train_time
        self.online_test_time = online_test_time
        self.start_test_time = online_test_time
        self.test_count = 0
        information = np.load(self._file("information.npy"))
        self.inter_all, self.user_number, self.item_number = information[0], information[1], information[2]
        print(information)
        users, items, total = set(), set(), 0
        for f in self.file_list:     # the reference prints (#interactions, #users, #items) of all train files
            a = self._load("train", f)
            total += a.shape[0]
            users.update(np.unique(a[:, 0]).tolist())
            items.update(np.unique(a[:, 1]).tolist())
        print(total, len(users), len(items))

    def _file(self, *parts):
        return os.path.join(self.path + self.dataname, *parts)

    def _load(self, split, name):
        return np.load(self._file(split, name + ".npy"))

    def reinit(self):
        self.test_count = 0
        self.start_test_time = copy.deepcopy(self.online_test_time)

    def _set_t(self, now):
        if self.MF_sample == "alone":
            return self._load("train", self.file_list[now])
        if self.MF_sample == "all":
            return self._load("test", self.file_list[now])
        raise TypeError("now such type when read next train sets")

    def _set_tt(self, now, label):
        src = now if self.current_as_set_tt else now + 1
        if self.TR_sample_type == "alone":
            p = self._file("train", self.file_list[src] + ".npy")
            print(label, p)
            return np.load(p)
        if self.TR_sample_type == "all":
            return self._load("test", self.file_list[src])
        raise TypeError("no such TR sample type")

    def next_train(self, d_time):
        """-> (set_t, set_tt, now_test, val) for stage d_time; (None,)*4 past the last period.
        Branches as the reference (data/dataset2.py:257-351): pure online training before
        online_test_time; afterwards test-then-train (or, with TR_stop_, test only)."""
        now = self.online_trian_time + d_time
        if now + 1 >= self.len:
            return None, None, None, None
        print("now time:", now)
        print("will be test data:", now + 1)
        if now + 1 < self.start_test_time:
            set_t = self._set_t(now)
            val = self._load("test", self.file_list[now + 1])
            print("now time:", self.file_list[now])
            set_tt = self._set_tt(now, "set_tt is:" if not self.current_as_set_tt else "set tt is:")
            if self.TR_sample_type == "all":
                print("t+1 data stes", self.file_list[now + 1])
            return set_t, set_tt, None, val
        if self.TR_stop_:
            set_t = self._set_t(now)
            print("now time:", self.file_list[now])
            print("will be test data:", self.test_list[self.test_count])
            now_test = self._load("test", self.test_list[self.test_count])
            self.test_count += 1
            return set_t, None, now_test, now_test
        set_t = self._set_t(now)
        val = self._load("test", self.file_list[now + 1])
        set_tt = self._set_tt(now, "settt is:")
        if self.TR_sample_type == "all":
            print("set_tt is", self.file_list[now + 1])
            print("t+1 datasets,", self.file_list[now + 1])
        now_test = self._load("test", self.test_list[self.test_count])
        print("real test:", self.test_list[self.test_count])
        self.test_count += 1
        return set_t, set_tt, now_test, val

"""Synthetic period data in the reference's on-disk format.

The reference ships no data (its datasets sit behind an external URL), so every
configuration in BASELINE.json runs on data written by this generator.  The
format is the one `transfer_data` reads (reference data/dataset2.py:229-232,
README.md:23-25):

    <root>/<name>/information.npy   int64 [3]  = [n_interactions, n_user, n_item]
    <root>/<name>/train/<p>.npy     int64 [n, 2]        (user, item)
    <root>/<name>/test/<p>.npy      int64 [n, 2 + neg]  (user, pos item, neg items...)

Users follow Zipf(a_user), items Zipf(a_item); negatives are drawn uniformly
and never equal the row's positive, so evaluation ranks are tie-free in the
sense the reference's top-k needs ("topk ties").
"""
import os

import numpy as np


def _zipf_probs(n, a):
    if a <= 0.0:
        return np.full(n, 1.0 / n)
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), a)
    return w / w.sum()


def sample_period(rng, n_inter, n_user, n_item, a_user=1.1, a_item=1.0, neg=999,
                  user_perm=None, item_perm=None):
    """One period: (train [n,2], test [n,2+neg]) int64 arrays."""
    pu = _zipf_probs(n_user, a_user)
    pi = _zipf_probs(n_item, a_item)
    users = rng.choice(n_user, size=n_inter, p=pu)
    items = rng.choice(n_item, size=n_inter, p=pi)
    if user_perm is not None:
        users = user_perm[users]
    if item_perm is not None:
        items = item_perm[items]
    train = np.stack([users, items], axis=1).astype(np.int64)
    # negatives: uniform over items, shifted past the positive so neg != pos
    negs = rng.randint(0, n_item - 1, size=(n_inter, neg)).astype(np.int64)
    negs += (negs >= items[:, None])
    test = np.concatenate([train, negs], axis=1)
    return train, test


def write_dataset(root, name, n_periods, n_inter, n_user, n_item, neg=999,
                  a_user=1.1, a_item=1.0, seed=2000):
    """Write an n_periods dataset under root/name/. Returns the information triple."""
    base = os.path.join(root, name)
    os.makedirs(os.path.join(base, "train"), exist_ok=True)
    os.makedirs(os.path.join(base, "test"), exist_ok=True)
    perm_rng = np.random.RandomState(seed - 1)
    user_perm = perm_rng.permutation(n_user)
    item_perm = perm_rng.permutation(n_item)
    total = 0
    for p in range(n_periods):
        rng = np.random.RandomState(seed + p)
        train, test = sample_period(rng, n_inter, n_user, n_item, a_user, a_item, neg,
                                    user_perm, item_perm)
        np.save(os.path.join(base, "train", "%d.npy" % p), train)
        np.save(os.path.join(base, "test", "%d.npy" % p), test)
        total += train.shape[0]
    info = np.array([total, n_user, n_item], dtype=np.int64)
    np.save(os.path.join(base, "information.npy"), info)
    return info


def synth_triples(rng, n, n_user, n_item, a_user=0.0, a_item=1.0):
    """(user, pos, neg) int64 triples for the bare embed+loss benchmark."""
    pu = _zipf_probs(n_user, a_user) if a_user > 0 else None
    users = rng.choice(n_user, size=n, p=pu) if pu is not None else rng.randint(0, n_user, size=n)
    if a_item > 0 and n_item <= (1 << 22):
        pi = _zipf_probs(n_item, a_item)
        pos = rng.choice(n_item, size=n, p=pi)
    else:
        pos = rng.randint(0, n_item, size=n)
    neg = rng.randint(0, n_item - 1, size=n)
    neg += (neg >= pos)
    return users.astype(np.int64), pos.astype(np.int64), neg.astype(np.int64)

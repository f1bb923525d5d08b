"""Batch pipeline and loaders around the hot path: the subset of
gnnflow/utils.py that feeds `TemporalSampler.sample` / `Cache.fetch_feature` — dataset and
feature loaders, negative samplers, `get_batch`, `build_dynamic_graph`, `prepare_input`,
`mfgs_to_cuda` — with the reference's names, arguments and outputs.

Root layout of a batch (gnnflow/utils.py:371-395, gnnflow/data.py:36-52):
    roots = [src(B) | dst(B) | neg_dst(B)] int64,  ts = [t | t | t] float32,  eid[B]
"""
import ctypes as C
import os
import weakref
from typing import List, Optional, Union

import numpy as np
import torch

from . import _capi
from .dynamic_graph import DynamicGraph

NODE_FEATS = None


def get_node_feats():
    return NODE_FEATS


def local_world_size() -> int:
    return int(os.environ["LOCAL_WORLD_SIZE"])


def local_rank() -> int:
    return int(os.environ["LOCAL_RANK"])


def rank() -> int:
    return torch.distributed.get_rank()


def get_project_root_dir() -> str:
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data_dir(data_dir):
    return os.path.join(get_project_root_dir(), "data") if data_dir is None else data_dir


# ---- datasets (gnnflow/utils.py:37-150) ---------------------------------------------
def load_dataset(dataset: str, data_dir: Optional[str] = None):
    """`<data_dir>/<dataset>/edges.csv` (columns src, dst, time, ext_roll; the unnamed first
    column is the edge id) -> (train, val, test, full) dataframes split where `ext_roll`
    turns 1 and 2.  The slices keep the file's row index (gnnflow/utils.py:37-75)."""
    import pandas as pd
    path = os.path.join(_data_dir(data_dir), dataset, 'edges.csv')
    if not os.path.exists(path):
        raise ValueError('{} does not exist'.format(path))
    full = pd.read_csv(path)
    full.rename(columns={'Unnamed: 0': 'eid'}, inplace=True)
    roll = full['ext_roll'].values
    train_end, val_end = roll.searchsorted(1), roll.searchsorted(2)
    return full[:train_end], full[train_end:val_end], full[val_end:], full


def load_dataset_in_chunks(dataset: str, data_dir: Optional[str] = None,
                           chunksize: int = 100000000):
    """Iterator over `edges.csv` in chunks of `chunksize` rows (gnnflow/utils.py:130-150)."""
    import pandas as pd
    path = os.path.join(_data_dir(data_dir), dataset, 'edges.csv')
    if not os.path.exists(path):
        raise ValueError('{} does not exist'.format(path))
    return pd.read_csv(path, chunksize=chunksize,
                       usecols=['src', 'dst', 'time', 'Unnamed: 0', 'ext_roll'])


def load_feat(dataset: str, data_dir: Optional[str] = None, shared_memory: bool = False,
              local_rank: int = 0, local_world_size: int = 1, memmap: bool = False,
              load_node: bool = True, load_edge: bool = True):
    """`node_features.npy` / `edge_features.npy` of a dataset -> (node_feats, edge_feats),
    either may be None but not both missing (gnnflow/utils.py:249-340).

    shared_memory=True is the reference's protocol for several ranks on one machine
    (:289-338): local rank 0 reads the files, converts to float32 and copies the tables into
    POSIX shared memory ('node_feats' / 'edge_feats', dgl.utils.shared_mem's role:
    gnnflow_amd.dgl_compat), broadcasts the shapes (torch.distributed object broadcast from
    rank 0 — a collective, as in the reference), the other local ranks map the same pages, and
    everybody meets at a barrier.  One host copy per machine instead of one per rank; what
    each MI355X rank then puts into HBM is its own business (Cache(feature_placement=...),
    or dist.FeatureShards.from_full for an owner shard)."""
    root = os.path.join(_data_dir(data_dir), dataset)
    node_path = os.path.join(root, 'node_features.npy')
    edge_path = os.path.join(root, 'edge_features.npy')
    if not os.path.exists(node_path) and not os.path.exists(edge_path):
        raise ValueError("Both {} and {} do not exist".format(node_path, edge_path))
    mode = "r+" if memmap else None

    def load(path, wanted):
        if not (wanted and os.path.exists(path)):
            return None
        arr = np.load(path, mmap_mode=mode, allow_pickle=False)
        return arr if memmap else torch.from_numpy(arr)

    node_feats = edge_feats = None
    if not shared_memory or local_rank == 0:
        node_feats, edge_feats = load(node_path, load_node), load(edge_path, load_edge)
    if not shared_memory:
        return node_feats, edge_feats

    def f32(x):
        return torch.as_tensor(np.asarray(x) if memmap else x).to(torch.float32)
    if local_world_size <= 1 and not torch.distributed.is_initialized():
        # one rank: nothing to share
        return (f32(node_feats) if node_feats is not None else None,
                f32(edge_feats) if edge_feats is not None else None)
    from .dgl_compat import create_shared_mem_array, get_shared_mem_array
    node_shm = edge_shm = None
    if local_rank == 0:
        if node_feats is not None:
            node_feats = f32(node_feats)
            node_shm = create_shared_mem_array('node_feats', node_feats.shape, node_feats.dtype)
            node_shm[:] = node_feats[:]
        if edge_feats is not None:
            edge_feats = f32(edge_feats)
            edge_shm = create_shared_mem_array('edge_feats', edge_feats.shape, edge_feats.dtype)
            edge_shm[:] = edge_feats[:]
        shapes = [tuple(node_feats.shape) if node_feats is not None else None,
                  tuple(edge_feats.shape) if edge_feats is not None else None]
        torch.distributed.broadcast_object_list(shapes, src=0)
    else:
        shapes = [None, None]
        torch.distributed.broadcast_object_list(shapes, src=0)
        if shapes[0] is not None:
            node_shm = get_shared_mem_array('node_feats', shapes[0], torch.float32)
        if shapes[1] is not None:
            edge_shm = get_shared_mem_array('edge_feats', shapes[1], torch.float32)
    torch.distributed.barrier()
    return node_shm, edge_shm


# ---- negative samplers (gnnflow/utils.py:343-366, 504-530) ----------------------------
class DstRandEdgeSampler:
    """Uniform negatives over the distinct destination ids seen so far."""

    def __init__(self, dst_list, seed=None):
        self.seed = None
        self.dst_list = np.unique(dst_list)
        if seed is not None:
            self.seed = seed
            self.random_state = np.random.RandomState(self.seed)

    def sample(self, size):
        draw = np.random.randint if self.seed is None else self.random_state.randint
        return self.dst_list[draw(0, len(self.dst_list), size)]

    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed)

    def add_dst_list(self, dst):
        self.dst_list = np.unique(np.concatenate((self.dst_list, dst)))


class RandEdgeSampler:
    """Uniform (src, dst) pairs over the distinct source / destination ids."""

    def __init__(self, src_list, dst_list, seed=None):
        self.seed = None
        self.src_list = np.unique(src_list)
        self.dst_list = np.unique(dst_list)
        if seed is not None:
            self.seed = seed
            self.random_state = np.random.RandomState(self.seed)

    def sample(self, size):
        draw = np.random.randint if self.seed is None else self.random_state.randint
        src_index = draw(0, len(self.src_list), size)
        dst_index = draw(0, len(self.dst_list), size)
        return self.src_list[src_index], self.dst_list[dst_index]

    def reset_random_state(self):
        self.random_state = np.random.RandomState(self.seed)


# ---- batches (gnnflow/utils.py:369-410) -----------------------------------------------
def _row_groups(df, batch_size, skip):
    """Row positions of every batch: rows `skip:` grouped by `index // batch_size` — the
    batch borders stay aligned to multiples of `batch_size` in *index* space, so a frame
    whose index does not start at a multiple (val / test slices, or a random start) opens
    with a short batch.  Groups come in ascending key order, rows keep their order."""
    keys = np.asarray(df.index // batch_size)[skip:]
    if len(keys) == 0:
        return
    order = np.argsort(keys, kind="stable")
    sorted_keys = keys[order]
    cuts = np.flatnonzero(sorted_keys[1:] != sorted_keys[:-1]) + 1
    for pos in np.split(order, cuts):
        yield pos + skip


def _random_start(num_chunks, batch_size, world_size):
    if num_chunks == 0:
        return 0
    device = "cuda:{}".format(local_rank()) if torch.cuda.is_available() else "cpu"
    randint = torch.randint(0, num_chunks, size=(1,), device=device)
    if world_size > 1:
        torch.distributed.broadcast(randint, src=0)
    return int(randint) * batch_size // num_chunks


def get_batch(df, batch_size: int, num_chunks: int, rand_edge_sampler: DstRandEdgeSampler,
              world_size: int = 1):
    """Yields (roots, ts, eid) per batch with one sampled negative destination per edge;
    `num_chunks` > 0 drops a random multiple of `batch_size / num_chunks` leading rows
    (same draw on every rank)."""
    skip = _random_start(num_chunks, batch_size, world_size)
    src, dst = df['src'].values, df['dst'].values
    time, eids = df['time'].values, df['eid'].values
    for pos in _row_groups(df, batch_size, skip):
        neg = rand_edge_sampler.sample(len(pos))
        t = time[pos]
        yield (np.concatenate([src[pos], dst[pos], neg]).astype(np.int64),
               np.concatenate([t, t, t]).astype(np.float32), eids[pos])


def get_batch_no_neg(df, batch_size: int):
    """As get_batch without negatives: roots = [src | dst]."""
    src, dst = df['src'].values, df['dst'].values
    time, eids = df['time'].values, df['eid'].values
    for pos in _row_groups(df, batch_size, 0):
        t = time[pos]
        yield (np.concatenate([src[pos], dst[pos]]).astype(np.int64),
               np.concatenate([t, t]).astype(np.float32), eids[pos])


# ---- graph construction (gnnflow/utils.py:413-462) -------------------------------------
def build_dynamic_graph(initial_pool_size: int, maximum_pool_size: int, mem_resource_type: str,
                        minimum_block_size: int, blocks_to_preallocate: int,
                        insertion_policy: str, undirected: bool, device: int = 0,
                        adaptive_block_size: bool = True, dataset_df=None,
                        *args, **kwargs) -> DynamicGraph:
    """DynamicGraph from a dataframe with columns src, dst, time, eid (or an empty one)."""
    if dataset_df is None:
        src = dst = ts = eids = None
    else:
        src = dataset_df['src'].values.astype(np.int64)
        dst = dataset_df['dst'].values.astype(np.int64)
        ts = dataset_df['time'].values.astype(np.float32)
        eids = dataset_df['eid'].values.astype(np.int64)
    return DynamicGraph(initial_pool_size, maximum_pool_size, mem_resource_type,
                        minimum_block_size, blocks_to_preallocate, insertion_policy,
                        src, dst, ts, eids, undirected, device, adaptive_block_size)


# ---- cache-free feature gather (gnnflow/utils.py:465-474) ------------------------------
_device_tables = {}


def _device_table(feats, device: torch.device) -> torch.Tensor:
    """The table as something the gather kernel can read: an fp32 device tensor (kept as is),
    a pinned fp32 host tensor (read over PCIe in place), or — for any other tensor / numpy
    array — a device copy made once and remembered for as long as the source lives."""
    if isinstance(feats, torch.Tensor) and feats.dtype == torch.float32 and feats.is_contiguous() \
            and (feats.is_cuda or feats.is_pinned()):
        return feats
    key = id(feats)
    hit = _device_tables.get(key)
    if hit is not None and hit[0]() is feats:
        return hit[1]
    t = feats if isinstance(feats, torch.Tensor) else torch.from_numpy(np.asarray(feats))
    table = t.to(device=device, dtype=torch.float32).contiguous()
    try:
        _device_tables[key] = (weakref.ref(feats, lambda _: _device_tables.pop(key, None)), table)
    except TypeError:
        pass
    return table


def _gather(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    lib = _capi.load()
    device = ids.device
    ids = ids.to(torch.int64).contiguous()
    out = torch.empty((ids.shape[0], table.shape[1]), dtype=torch.float32, device=device)
    if ids.shape[0]:
        with torch.cuda.device(device):
            _capi.check(lib.gf_gather_rows(
                table.data_ptr(), table.shape[0], table.shape[1], ids.data_ptr(), ids.shape[0],
                out.data_ptr(), device.index,
                C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return out


def prepare_input(mfgs, node_feats, edge_feats):
    """`srcdata['h'] = node_feats[ID].float()` for the blocks of mfgs[0] and
    `edata['f'] = edge_feats[eID].float()` for every block — the cache-free fetch, on the
    HIP gather kernel."""
    if node_feats is not None:
        for b in mfgs[0]:
            ids = b.srcdata['ID']
            b.srcdata['h'] = _gather(_device_table(node_feats, ids.device), ids)
    if edge_feats is not None:
        for mfg in mfgs:
            for b in mfg:
                ids = b.edata['ID']
                b.edata['f'] = _gather(_device_table(edge_feats, ids.device), ids)
    return mfgs


def mfgs_to_cuda(mfgs: List[List], device: Union[str, torch.device]):
    """The sampler's blocks already live on its GPU, so this only moves blocks that were built
    elsewhere (gnnflow/utils.py:477-481)."""
    for mfg in mfgs:
        for i in range(len(mfg)):
            mfg[i] = mfg[i].to(device)
    return mfgs


def get_pinned_buffers(fanouts, sample_history, batch_size, dim_node, dim_edge):
    """The reference stages cache misses through pinned host buffers
    (gnnflow/utils.py:484-501); the HIP gather reads the feature tables directly, so none are
    needed — empty lists keep `Cache(...)` call sites unchanged."""
    return [], []


class EarlyStopMonitor:
    """Stops training when the monitored value has not improved (relatively, by more than
    `tolerance`) for `max_round` consecutive epochs (gnnflow/utils.py:532-561; imported by
    scripts/offline_edge_prediction.py)."""

    def __init__(self, max_round=5, higher_better=True, tolerance=1e-10):
        self.max_round = max_round
        self.num_round = 0
        self.epoch_count = 0
        self.best_epoch = 0
        self.last_best = None
        self.higher_better = higher_better
        self.tolerance = tolerance

    def early_stop_check(self, curr_val):
        value = curr_val if self.higher_better else -curr_val
        if self.last_best is None:
            self.last_best = value
        elif (value - self.last_best) / np.abs(self.last_best) > self.tolerance:
            self.last_best = value
            self.num_round = 0
            self.best_epoch = self.epoch_count
        else:
            self.num_round += 1
        self.epoch_count += 1
        return self.num_round >= self.max_round


def bind_to_device_cpus(device=0):
    """Restricts the calling process to the CPUs of the GPU's NUMA node (what one would do
    with `numactl --cpunodebind` per rank).  Every kernel launch is a few posted writes to the
    device's queue and doorbell; issued from the other socket each costs about a microsecond
    more, and a batch-600 step is 8 launches: measured 36-41 us per step from the GPU's own
    node against 39-52 us from the remote one on a 2-socket MI355X host.  Threads created
    afterwards (the library's enqueue thread) inherit the mask.  Returns the CPU set, or None
    when the topology cannot be read (then nothing is changed)."""
    import os
    try:
        import torch
        index = torch.device(device).index if not isinstance(device, int) else device
        p = torch.cuda.get_device_properties(index if index is not None else 0)
        bus = "{:04x}:{:02x}:{:02x}.0".format(getattr(p, "pci_domain_id", 0), p.pci_bus_id,
                                              p.pci_device_id)
        with open("/sys/bus/pci/devices/{}/local_cpulist".format(bus)) as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        mask = cpus & os.sched_getaffinity(0)
        if not mask:
            return None
        os.sched_setaffinity(0, mask)
        return mask
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None

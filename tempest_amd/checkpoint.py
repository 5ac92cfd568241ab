"""Native checkpoint format: per-shard raw SoA dumps + one JSON iteration table.

The reference persists a dill pickle of {_current, _history, n_dim, ...} (core.py:249-279,
state_manager.py:505-652) and, on load, restores the current state only (core.py:289).  That layout is still read and
written for small single-GPU runs (`*.state`).  A long sharded run needs something else: every rank holds gigabytes of
history in the dimension-major layout the kernels use, so a checkpoint is a directory

    <name>.ckpt/
        meta.json                 format/version, n_dim, world size, the iteration table (beta_t, logZ_t, rows per
                                  shard and globally), every scalar history, the current scalars, the counter-based RNG
                                  state (seed, tick), n_total
        shard0000.hist_u.f64      float64 little-endian, [n_dim][rows of this shard]  (the device layout, as is)
        shard0000.hist_x.f64
        shard0000.hist_logl.f64   [rows]
        shard0000.cur_u.f64       current active set of the shard, [n_dim][n]
        shard0000.cur_x.f64, shard0000.cur_logl.f64, shard0000.cur_assign.i32
        shard0001....             one set per rank; written and read by that rank only (no gather)

written into `<name>.ckpt.tmp/` and renamed once every shard is complete.  The cached log-mixture is not stored: it is
a function of logl and the iteration table and is rebuilt on load by the same kernel that maintains it.
"""
import json
import os
import shutil
from pathlib import Path

import numpy as np

FORMAT = "tempest_amd-checkpoint"
VERSION = 1
SUFFIX = ".ckpt"


def _old(path) -> Path:
    path = Path(path)
    return path.with_name(path.name + ".old")


def _resolve(path) -> Path:
    """The directory holding the newest COMPLETE checkpoint under this name: `path`, or -- if a writer died between the two
    renames of save() -- the previous checkpoint it had just moved aside to `<path>.old`."""
    p = Path(path)
    if p.is_dir() and (p / "meta.json").exists():
        return p
    o = _old(p)
    if o.is_dir() and (o / "meta.json").exists():
        return o
    return p


def is_native(path) -> bool:
    p = _resolve(path)
    return p.is_dir() and (p / "meta.json").exists()


def _fsync_path(p):
    """Flush a file (or a directory's entries) to stable storage; best effort on file systems that refuse it."""
    try:
        fd = os.open(str(p), os.O_RDONLY)
    except OSError:
        return
    try:
        os.fsync(fd)
    except OSError:
        pass
    finally:
        os.close(fd)


def wants_native(path, fmt=None, comm=None) -> bool:
    if fmt is not None:
        if fmt not in ("native", "dill"):
            raise ValueError(f"checkpoint format must be 'native' or 'dill', got {fmt!r}")
        return fmt == "native"
    if comm is not None and comm.active:
        return True                      # a pickle per rank under one name would collide: shards go to a directory
    return Path(path).suffix == SUFFIX or Path(path).is_dir()


def _json_scalar(v):
    if v is None:
        return None
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return float(v)
    if isinstance(v, np.ndarray):
        return {"__ndarray__": v.tolist(), "dtype": str(v.dtype)}
    raise TypeError(f"cannot store {type(v).__name__} in a checkpoint")


def _from_json_scalar(v):
    if isinstance(v, dict) and "__ndarray__" in v:
        return np.array(v["__ndarray__"], dtype=v["dtype"])
    return v


def _barrier(comm):
    if comm is not None and comm.active:
        comm.barrier()


def save(core, path):
    """Write the run held by `core` (SamplerCore) to the directory `path`."""
    import torch
    from .device import KEY_LOGL, KEY_U, KEY_X
    st = core.state
    comm = st.comm
    active = comm is not None and comm.active
    rank = comm.rank if active else 0
    world = comm.world_size if active else 1
    path = Path(path)
    tmp = path.with_name(path.name + ".tmp")
    if rank == 0:
        old = _old(path)
        if not path.exists() and old.exists():
            os.rename(old, path)         # an earlier writer died between its two renames: its previous checkpoint is the good one
        if tmp.exists():
            shutil.rmtree(tmp)
        tmp.mkdir(parents=True)
    _barrier(comm)
    ctx = st.ctx
    ctx.use_current_stream()
    ctx.synchronize()
    stem = tmp / f"shard{rank:04d}"
    size = ctx.size
    for name, key in (("hist_u", KEY_U), ("hist_x", KEY_X), ("hist_logl", KEY_LOGL)):
        arr = ctx.history_read(key, soa=True) if size else np.empty(0)
        np.ascontiguousarray(arr, dtype="<f8").tofile(f"{stem}.{name}.f64")
    cur_n = 0
    for name in ("u", "x", "logl"):
        t = st.dev(name)
        if t is not None:
            a = t.detach().cpu().numpy()
            cur_n = a.shape[-1]
            np.ascontiguousarray(a, dtype="<f8").tofile(f"{stem}.cur_{name}.f64")
    t = st.dev("assignments")
    if t is not None:
        np.ascontiguousarray(t.detach().to(torch.int32).cpu().numpy(), dtype="<i4").tofile(f"{stem}.cur_assign.i32")
    if st._blobs or st._current.get("blobs") is not None:
        np.save(f"{stem}.blobs.npy", np.array([st._blobs, st._current.get("blobs")], dtype=object), allow_pickle=True)
    for f in tmp.glob(f"shard{rank:04d}.*"):      # this rank's shard files reach the disk before the directory is renamed
        _fsync_path(f)
    _barrier(comm)
    if rank == 0:
        cur = {k: _json_scalar(v) for k, v in st._current.items()
               if k not in ("u", "x", "logl", "assignments", "blobs")}
        meta = {
            "format": FORMAT, "version": VERSION, "n_dim": st.n_dim, "world_size": world,
            "n_local_t": [int(v) for v in st._n_local], "n_global_t": [int(v) for v in st._n_global],
            "scalars": {k: [_json_scalar(v) for v in vals] for k, vals in st._scalars.items()},
            "current": cur, "current_rows_per_shard": int(cur_n),
            "rng": [int(core.rng.seed), int(core.rng.tick)],
            "n_total": _json_scalar(getattr(core, "n_total", None)),
            "random_state": _json_scalar(core.config.random_state),
            "logz_err": _json_scalar(getattr(core, "logz_err", None)),
            "sampler": {"n_particles": core.config.n_particles, "sample": core.config.sample,
                        "resample": core.config.resample, "clustering": bool(core.config.clustering)},
        }
        with open(tmp / "meta.json", "w") as f:
            json.dump(meta, f, indent=1)
            f.flush()
            os.fsync(f.fileno())
        _fsync_path(tmp)
        # replace: old -> .old, new -> path, then drop .old.  Between the two renames the complete checkpoint is `.old`:
        # load() / is_native() fall back to it, and the next save() moves it back before it touches anything.
        old = _old(path)
        if path.exists():
            if old.exists():
                shutil.rmtree(old)
            os.rename(path, old)
        os.rename(tmp, path)
        _fsync_path(path.parent)
        if old.exists():
            shutil.rmtree(old)
    _barrier(comm)


def load(core, path):
    """Restore history, current state and RNG position from a checkpoint directory (same world size)."""
    import torch
    st = core.state
    comm = st.comm
    active = comm is not None and comm.active
    rank = comm.rank if active else 0
    world = comm.world_size if active else 1
    path = _resolve(path)
    with open(path / "meta.json") as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT:
        raise ValueError(f"{path} is not a {FORMAT} directory")
    if meta["version"] > VERSION:
        raise ValueError(f"checkpoint version {meta['version']} is newer than this build ({VERSION})")
    if meta["world_size"] != world:
        raise ValueError(f"checkpoint was written by {meta['world_size']} rank(s), this run has {world}: "
                         "shards are restored one per rank")
    if meta["n_dim"] != st.n_dim:
        raise ValueError(f"checkpoint has n_dim={meta['n_dim']}, sampler has {st.n_dim}")
    d = st.n_dim
    stem = path / f"shard{rank:04d}"
    n_t, n_g = meta["n_local_t"], meta["n_global_t"]
    size = int(np.sum(n_t)) if n_t else 0
    ctx = st.ctx
    ctx.use_current_stream()
    st._scalars = {k: [_from_json_scalar(v) for v in vals] for k, vals in meta["scalars"].items()}
    if size:
        rd = lambda name, shape: np.fromfile(f"{stem}.{name}.f64", dtype="<f8").reshape(shape)  # noqa: E731
        T = len(n_t)
        beta = [float(v) for v in st._scalars["beta"][:T]]
        logz = [float(v) for v in st._scalars["logz"][:T]] if len(st._scalars["logz"]) >= T else [0.0] * T
        ctx.history_load(rd("hist_u", (d, size)), rd("hist_x", (d, size)), rd("hist_logl", (size,)), beta, logz, n_t,
                         n_t_global=n_g, soa=True)
    else:
        ctx.history_clear()
    st._n_local, st._n_global = [int(v) for v in n_t], [int(v) for v in n_g]
    for k, v in meta["current"].items():
        st._current[k] = _from_json_scalar(v)
    n = int(meta.get("current_rows_per_shard", 0))
    dev = st.device
    for name in ("u", "x", "logl"):
        f = Path(f"{stem}.cur_{name}.f64")
        if f.exists() and n:
            a = np.fromfile(f, dtype="<f8").reshape((d, n) if name != "logl" else (n,))
            st._current[name] = torch.from_numpy(a).to(dev)
        else:
            st._current[name] = None
    f = Path(f"{stem}.cur_assign.i32")
    st._current["assignments"] = torch.from_numpy(np.fromfile(f, dtype="<i4")).to(dev) if f.exists() else None
    f = Path(f"{stem}.blobs.npy")
    if f.exists():
        blobs = np.load(f, allow_pickle=True)
        st._blobs, st._current["blobs"] = list(blobs[0]), blobs[1]
    st._invalidate_cache()
    core.rng.seed, core.rng.tick = int(meta["rng"][0]), int(meta["rng"][1])
    if meta.get("n_total") is not None:
        core.n_total = meta["n_total"]
    core.logz_err = meta.get("logz_err")
    return meta

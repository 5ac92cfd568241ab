"""Device-resident StateManager with the reference's interface (tempest/state_manager.py).

* The *current* particle arrays (u, x as (n_dim, n) SoA tensors; logl; assignments) live on the GPU;
  `get_current` hands out NumPy copies in the reference's (n, n_dim) layout and `set_current`
  accepts them, so step-level code written against the reference keeps working.
* The *history* of u, x, logl is the ctx-owned persistent ensemble in HBM (SoA, see csrc/ctx.hip)
  together with the cached log-mixture; per-iteration scalars stay in host lists like the reference.
* `compute_logw_and_logz` (state_manager.py:418-480) is one streaming reduction on the device.
"""
import os
from pathlib import Path
from typing import Optional, Union

import numpy as np

CURRENT_STATE_KEYS = frozenset({
    "u", "x", "logl", "assignments", "blobs", "acceptance", "steps", "efficiency", "ess", "cv", "beta", "logz",
    "calls", "iter",
})
HISTORY_STATE_KEYS = frozenset({
    "u", "x", "logl", "blobs", "iter", "logz", "calls", "steps", "efficiency", "ess", "cv", "acceptance", "beta",
})
REQUIRED_COMMIT_KEYS = frozenset({"beta", "logl"})

_DEVICE_ARRAYS = ("u", "x", "logl")
_SCALAR_HISTORY = tuple(sorted(HISTORY_STATE_KEYS - set(_DEVICE_ARRAYS) - {"blobs"}))


class StateManager:
    """Current state + persistent history of a Persistent Sampling run, backed by one GPU context."""

    def __init__(self, n_dim: int, device=None, comm=None, capacity_hint: int = 0):
        self.n_dim = n_dim
        self._device_arg = device
        self._capacity_hint = capacity_hint
        import threading
        self._ctx_lock = threading.Lock()
        self._ctx = None
        self.comm = comm
        self._current = dict.fromkeys(CURRENT_STATE_KEYS, None)
        self._scalars = {k: [] for k in _SCALAR_HISTORY}
        self._blobs = []
        self._n_local = []       # rows per committed iteration held by this rank
        self._n_global = []
        self._results_dict = None

    # ------------------------------------------------------------------------------ device
    @property
    def ctx(self):
        if self._ctx is None:
            with self._ctx_lock:          # the start-up warm-up (_warm.py) may be creating it in its own thread
                if self._ctx is None:
                    from .device import HipContext
                    ctx = HipContext(self.n_dim, self._device_arg, self._clamped_hint())
                    if self.comm is not None and self.comm.active:
                        # the library issues its own small collectives (reweight triples, global trim / fit / cumulative weights)
                        self.comm.attach(ctx)
                    self._ctx = ctx
        return self._ctx

    def _clamped_hint(self) -> int:
        """History rows to reserve up front: the caller's hint, but never more than a quarter of the free device memory
        (a history that outgrows it falls back to geometric growth, whose reallocations stall the stream)."""
        rows = int(self._capacity_hint or 0)
        if rows <= 0:
            return 0
        try:
            import torch
            dev = self._device_arg if self._device_arg is not None else torch.cuda.current_device()
            free, _ = torch.cuda.mem_get_info(dev)
            rows = min(rows, int(0.25 * free) // ((2 * self.n_dim + 2) * 8))
        except Exception:
            pass
        return max(rows, 0)

    @property
    def device(self):
        return self.ctx.device

    def _to_device(self, key, value):
        import torch
        if isinstance(value, torch.Tensor):
            return value
        a = np.asarray(value)
        if key in ("u", "x"):
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, self.n_dim).T)
            return torch.from_numpy(a).to(self.device)
        if key == "logl":
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)).to(self.device)
        if key == "assignments":
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32).reshape(-1)).to(self.device)
        raise KeyError(key)

    @staticmethod
    def _to_host(key, value):
        import torch
        if not isinstance(value, torch.Tensor):
            return value.copy() if isinstance(value, np.ndarray) else value
        a = value.detach().cpu().numpy()
        if key in ("u", "x"):
            return np.ascontiguousarray(a.T)
        if key == "assignments":
            return a.astype(np.int64)
        return a.copy()

    def dev(self, key):
        """Device tensor of a current array (no copy): u, x as (n_dim, n); logl (n,); assignments int32."""
        return self._current[key]

    # ------------------------------------------------------------------------- current state
    def get_current(self, key: Optional[str] = None):
        if key is None:
            return {k: self._to_host(k, v) for k, v in self._current.items()}
        self._validate_current_key(key)
        return self._to_host(key, self._current[key])

    def set_current(self, key: str, value, copy: bool = True):
        self._validate_current_key(key)
        self._store(key, value, copy)
        self._invalidate_cache()

    def update_current(self, data_dict: dict, copy: bool = True):
        for key in data_dict:
            self._validate_current_key(key)
        for key, value in data_dict.items():
            self._store(key, value, copy)
        self._invalidate_cache()

    def _store(self, key, value, copy):
        import torch
        if value is None:
            self._current[key] = None
        elif key in ("u", "x", "logl", "assignments"):
            if isinstance(value, torch.Tensor):
                self._current[key] = value.clone() if copy else value
            else:
                self._current[key] = self._to_device(key, value)
        elif isinstance(value, np.ndarray):
            self._current[key] = value.copy() if copy else value
        else:
            self._current[key] = value

    # ------------------------------------------------------------------------------ history
    def get_history_length(self) -> int:
        return len(self._scalars["beta"])

    def _offsets(self):
        return np.concatenate([[0], np.cumsum(self._n_local)]).astype(np.int64)

    def get_history(self, key: str, index: Optional[int] = None, flat: bool = False):
        self._validate_history_key(key)
        from .device import KEY_LOGL, KEY_U, KEY_X
        dkey = {"u": KEY_U, "x": KEY_X, "logl": KEY_LOGL}.get(key)
        if index is not None:
            n_it = len(self._n_local) if dkey is not None else (len(self._blobs) if key == "blobs" else len(self._scalars[key]))
            if index >= n_it or index < 0:
                raise IndexError(f"Index {index} out of range for history key '{key}'")
            if dkey is not None:
                off = self._offsets()
                return self.ctx.history_read(dkey, int(off[index]), int(self._n_local[index]))
            if key == "blobs":
                return np.array(self._blobs[index], copy=True)
            v = self._scalars[key][index]
            return v.copy() if isinstance(v, np.ndarray) else v
        if dkey is not None:
            if len(self._n_local) == 0:
                if flat:
                    raise ValueError("need at least one array to concatenate")
                return np.array([])
            full = self.ctx.history_read(dkey)
            if flat:
                return full
            off = self._offsets()
            return np.array([full[off[t]:off[t + 1]] for t in range(len(self._n_local))])
        if key == "blobs":
            return np.concatenate(self._blobs) if flat else np.array(self._blobs)
        vals = self._scalars[key]
        return np.concatenate(vals) if flat else np.array(vals)

    def get_last_history(self, key: str, default=None):
        self._validate_history_key(key)
        n = self.get_history_length() if key not in _DEVICE_ARRAYS else len(self._n_local)
        if key == "blobs":
            n = len(self._blobs)
        elif key in self._scalars:
            n = len(self._scalars[key])
        if n == 0:
            return default
        return self.get_history(key, index=n - 1)

    def commit_current_to_history(self, strict: bool = False):
        """Append the current state to the history (state_manager.py:356-416).  The array keys go to the
        device ensemble, which also folds the new iteration into the cached log-mixture."""
        if strict:
            missing = [k for k in REQUIRED_COMMIT_KEYS if self._current.get(k) is None]
            if missing:
                raise ValueError(
                    f"Strict mode enabled: required keys are missing or None: {sorted(missing)}. "
                    f"Required keys: {sorted(REQUIRED_COMMIT_KEYS)}")
        logl = self._current["logl"]
        if logl is not None:
            import torch
            beta, logz = self._current["beta"], self._current["logz"]
            if beta is None:
                raise ValueError("cannot commit particle arrays without 'beta' (it defines the mixture term)")
            n = int(logl.shape[0])
            u, x = self._current["u"], self._current["x"]
            if u is None:
                u = torch.zeros(self.n_dim, n, dtype=torch.float64, device=self.device)
            if x is None:
                x = torch.zeros(self.n_dim, n, dtype=torch.float64, device=self.device)
            # shards are equal by construction (n_particles divisible by the number of ranks): no collective needed
            n_glob = n if self.comm is None else n * self.comm.world_size
            self.ctx.use_current_stream()
            self.ctx.history_append(u.contiguous(), x.contiguous(), logl.contiguous(), float(beta),
                                    0.0 if logz is None else float(logz), n_glob)
            self._n_local.append(n)
            self._n_global.append(n_glob)
        for k in _SCALAR_HISTORY:
            v = self._current[k]
            if v is not None:
                self._scalars[k].append(v.copy() if isinstance(v, np.ndarray) else v)
        if self._current["blobs"] is not None:
            self._blobs.append(np.array(self._current["blobs"], copy=True))
        self._invalidate_cache()

    # ---------------------------------------------------------------------- weights / evidence
    def n_history_global(self) -> int:
        return int(np.sum(self._n_global)) if self._n_global else 0

    def reweight_eval(self, betas):
        """Global (vmax, s1, s2) per trial beta: one device pass; with a communicator attached the library all-gathers
        the ranks' triples and merges them on the device, and the result arrives through the same pinned mailbox."""
        ctx = self.ctx
        ctx.use_current_stream()
        return ctx.reweight_eval(betas)

    def compute_logw_and_logz(self, beta_final: float = 1.0, normalize: bool = True):
        """Importance log-weights of every stored particle for the target at beta_final and the
        log-evidence estimate (state_manager.py:418-480)."""
        if self.get_history_length() == 0 or len(self._n_local) == 0:
            return np.array([]), -np.inf
        if len(self._n_local) != self.get_history_length():
            raise ValueError("history of 'beta' and of the particle arrays differ in length")
        m, s1, _ = self.reweight_eval([beta_final])[0]
        logz = float(m + np.log(s1))
        nh = self.n_history_global()
        logw = self.ctx.logw(beta_final, nh).cpu().numpy()
        if normalize:
            logw = logw - (logz + np.log(nh))
        return logw, logz

    def compute_results(self) -> dict:
        if self._results_dict is None:
            out = {}
            for key in HISTORY_STATE_KEYS:
                if key == "blobs" and not self._blobs:
                    out[key] = np.array([])
                    continue
                out[key] = self.get_history(key)
            out["logw"], _ = self.compute_logw_and_logz(1.0)
            self._results_dict = out
        return self._results_dict

    # --------------------------------------------------------------------------- persistence
    def to_dict(self) -> dict:
        """Same {_current, _history, n_dim} layout as the reference (state_manager.py:505-531), host arrays."""
        hist = {k: list(v) for k, v in self._scalars.items()}
        for key in _DEVICE_ARRAYS:
            hist[key] = [self.get_history(key, index=t) for t in range(len(self._n_local))]
        hist["blobs"] = list(self._blobs)
        return {"_current": self.get_current(), "_history": hist, "n_dim": self.n_dim}

    @classmethod
    def from_dict(cls, state_dict: dict) -> "StateManager":
        inst = cls(state_dict.get("n_dim", 1))
        inst.update_from_dict(state_dict)
        return inst

    def update_from_dict(self, state_dict: dict):
        if "n_dim" in state_dict and state_dict["n_dim"] != self.n_dim:
            self.n_dim = state_dict["n_dim"]
            self._ctx = None
        if "_current" in state_dict:
            for k, v in state_dict["_current"].items():
                if k in CURRENT_STATE_KEYS:
                    self._store(k, v, True)
        if "_history" in state_dict:
            h = state_dict["_history"]
            for k in _SCALAR_HISTORY:
                if k in h:
                    self._scalars[k] = list(h[k])
            if "blobs" in h:
                self._blobs = list(h["blobs"])
            if "logl" in h and len(h["logl"]):
                logl = [np.asarray(a, dtype=np.float64) for a in h["logl"]]
                T = len(logl)
                n_t = [a.size for a in logl]
                cat = lambda key: (np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1, self.n_dim)  # noqa: E731
                                                   for a in h[key]]) if key in h and len(h[key]) == T else None)
                beta = list(self._scalars["beta"])[:T]
                logz = list(self._scalars["logz"])[:T] if len(self._scalars["logz"]) >= T else [0.0] * T
                self.ctx.history_load(cat("u"), cat("x"), np.concatenate(logl), beta, logz, n_t)
                self._n_local = list(n_t)
                self._n_global = list(n_t)
            elif "logl" in h:
                self.ctx.history_clear()
                self._n_local, self._n_global = [], []
        self._invalidate_cache()

    def save_state(self, path: Union[str, Path], exclude: Optional[list] = None):
        """Atomic dill dump of to_dict() (state_manager.py:597-633)."""
        import dill
        print(f"Saving state to {path}")
        Path(path).parent.mkdir(exist_ok=True)
        temp_path = Path(path).with_suffix(".temp")
        d = self.to_dict()
        for key in (exclude if exclude is not None else ["pbar", "pool", "distribute"]):
            d.pop(key, None)
        with open(temp_path, "wb") as f:
            dill.dump(file=f, obj=d)
            f.flush()
            os.fsync(f.fileno())
        os.rename(temp_path, path)

    def load_state(self, path: Union[str, Path]):
        import dill
        with open(path, "rb") as f:
            d = dill.load(file=f)
        self.update_from_dict(d)

    # ------------------------------------------------------------------------------ helpers
    def _validate_current_key(self, key: str):
        if key not in CURRENT_STATE_KEYS:
            raise ValueError(f"Invalid current state key '{key}'. Valid keys: {sorted(CURRENT_STATE_KEYS)}")

    def _validate_history_key(self, key: str):
        if key not in HISTORY_STATE_KEYS:
            raise ValueError(f"Invalid history state key '{key}'. Valid keys: {sorted(HISTORY_STATE_KEYS)}")

    def _invalidate_cache(self):
        self._results_dict = None

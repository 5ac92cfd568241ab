"""User callbacks as HIP device functions, compiled into the MCMC step.

    cb = tempest_amd.HipCallbacks(n_dim=10, source='''
        __device__ void prior_transform(const double* u, double* x) {
          for (int j = 0; j < N_DIM; ++j) x[j] = 20.0 * u[j] - 10.0;
        }
        __device__ double log_likelihood(const double* x) {
          double s = 0.0;
          for (int j = 0; j < N_DIM; j += 2) {
            double a = x[j] * x[j] - x[j + 1], b = x[j] - 1.0;
            s += 10.0 * a * a + b * b;
          }
          return -s;
        }''')
    sampler = tempest_amd.Sampler(cb.prior_transform, cb.log_likelihood, 10, vectorize=True, ...)

`cb.prior_transform` / `cb.log_likelihood` are ordinary vectorised callbacks (torch-ROCm tensors or NumPy arrays in,
the same kind out), so everything that calls them generically keeps working; when a Sampler is given BOTH from the same
object, the mutation step skips them and launches the plugin's fused kernel instead (proposal -> [x' = prior(u'),
l' = loglike(x'), Metropolis update] -> adaptation: three launches per step instead of a chain of elementwise ones).

The source is compiled with hipcc for gfx950 into a shared library cached by content hash (in-tree under
tempest_amd/_plugins/ when writable, else ~/.cache/tempest_amd/plugins).  The reference has no counterpart: its
callbacks are Python functions evaluated on the host (mcmc.py:152-160, core.py:317-358).
"""
import ctypes as C
import hashlib
import os
import shutil
import subprocess
import tempfile
from pathlib import Path

import numpy as np

from ._lib import TempestHipError

_CSRC = Path(__file__).resolve().parent / "csrc"
_TEMPLATE = _CSRC / "user_plugin.hip.in"
_ARCH = "gfx950"
_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={_ARCH}", "-ffp-contract=on", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise TempestHipError("HipCallbacks needs hipcc (set HIPCC or install ROCm) to compile the user source")


def _cache_dirs():
    yield Path(__file__).resolve().parent / "_plugins"
    yield Path(os.environ.get("XDG_CACHE_HOME", Path.home() / ".cache")) / "tempest_amd" / "plugins"
    yield Path(tempfile.gettempdir()) / "tempest_amd_plugins"


def plugin_source(source: str) -> str:
    return _TEMPLATE.read_text().replace("@USER_SOURCE@", source)


_TOOLCHAIN = None


def _toolchain_id() -> str:
    """What identifies the compiler for the cache key: `hipcc --version` (a plugin shares struct layouts and inlined device
    code with libtempest_hip: a cached object from another toolchain must not be picked up silently)."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            r = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60)
            _TOOLCHAIN = (r.stdout + r.stderr).strip() or "unknown"
        except Exception:
            _TOOLCHAIN = "unknown"
    return _TOOLCHAIN


def build_plugin(source: str, n_dim: int, verbose: bool = False) -> Path:
    """Compile (or find in the cache) the plugin for `source`; returns the path of the shared library."""
    text = plugin_source(source)
    deps = (_CSRC / "common.h").read_bytes() + (_CSRC.parent.parent / "include" / "tempest_hip.h").read_bytes()
    key = f"|{n_dim}|{_ARCH}|{' '.join(_FLAGS)}|{_toolchain_id()}"
    tag = hashlib.sha256(text.encode() + deps + key.encode()).hexdigest()[:20]
    name = f"tphu_{n_dim}d_{tag}.so"
    for d in _cache_dirs():
        if (d / name).exists():
            return d / name
    last = None
    for d in _cache_dirs():
        try:
            d.mkdir(parents=True, exist_ok=True)
            with tempfile.TemporaryDirectory(dir=d) as tmp:
                src = Path(tmp) / "plugin.hip"
                src.write_text(text)
                out = Path(tmp) / name
                cmd = [_hipcc(), *_FLAGS, f"-DN_DIM={int(n_dim)}", f"-I{_CSRC}", str(src), "-o", str(out)]
                if verbose:
                    print(" ".join(cmd))
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise TempestHipError("HipCallbacks: hipcc failed\n" + r.stderr[-4000:])
                os.replace(out, d / name)        # atomic: concurrent ranks compile the same hash to the same name
            return d / name
        except OSError as e:                     # read-only location: try the next one
            last = e
    raise TempestHipError(f"HipCallbacks: no writable plugin cache directory ({last})")


class HipCallbacks:
    """prior_transform + log_likelihood as HIP device functions (see the module docstring)."""

    def __init__(self, source: str, n_dim: int, fused: bool = True, verbose: bool = False, whole_step: bool = True,
                 persistent: bool = False):
        if not isinstance(n_dim, int) or n_dim <= 0:
            raise ValueError(f"n_dim must be a positive int, got {n_dim!r}")
        for fn in ("prior_transform", "log_likelihood"):
            if fn not in source:
                raise ValueError(f"HipCallbacks source must define __device__ {fn}(...)")
        self.n_dim, self.source, self.fused = n_dim, source, bool(fused)
        self.whole_step = whole_step           # False: proposal and evaluate+accept as two kernels; "always": at any size
        # A whole run of steps in ONE cooperative launch (tphu_run) where the whole-step kernel applies.  OFF by default: measured
        # at 131 072 particles (profiles/r04_persistent_run.json) a step costs 32.5 us inside that launch against 30.0 us step by
        # step under the captured hipGraph -- 23.0 us whole-step kernel + 6.8 us tph_adapt + 0.4 us between them; the grid
        # barrier plus every workgroup's own sum of the tile partials cost more than the adaptation launch they replace.
        # TEMPEST_AMD_PERSISTENT=1/0 overrides the argument.
        env = os.environ.get("TEMPEST_AMD_PERSISTENT")
        self.persistent = bool(persistent) if env is None else env != "0"
        self.run_groups = 0                    # > 0 limits the workgroups of that launch (tests: several tiles per workgroup)
        self.path = build_plugin(source, n_dim, verbose)
        import torch  # noqa: F401  (its HIP runtime must be the one in the process, as for libtempest_hip)
        lib = C.CDLL(str(self.path))
        ptr, i64 = C.c_void_p, C.c_int64
        lib.tphu_last_error.restype = C.c_char_p
        lib.tphu_n_dim.restype = C.c_int
        lib.tphu_prior.argtypes = [ptr, ptr, i64, i64, ptr, i64]
        lib.tphu_like.argtypes = [ptr, ptr, i64, i64, ptr]
        lib.tphu_accept.argtypes = [ptr, C.c_int, C.c_double, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, C.c_int, ptr,
                                    C.c_uint64, C.c_uint32, i64, ptr, ptr, ptr, ptr]
        lib.tphu_step.argtypes = [ptr, C.c_int, C.c_double, ptr, ptr, ptr, i64, i64, ptr, ptr, ptr, ptr, ptr, ptr, C.c_uint64,
                                  C.c_uint32, C.c_uint32, i64, ptr, ptr, C.c_int]
        lib.tphu_run.argtypes = [ptr, C.c_int, ptr, ptr, ptr, i64, i64, ptr, ptr, ptr, ptr, ptr, ptr, C.c_uint64, C.c_uint32, C.c_uint32,
                                 i64, ptr, ptr, ptr, ptr, C.c_double, C.c_int, C.c_int, ptr, C.c_int, C.c_int, C.c_int, C.c_int]
        for f in (lib.tphu_prior, lib.tphu_like, lib.tphu_accept, lib.tphu_step, lib.tphu_run):
            f.restype = C.c_int
        if lib.tphu_n_dim() != n_dim:
            raise TempestHipError(f"plugin {self.path} was built for n_dim={lib.tphu_n_dim()}")
        self.lib = lib

    # ------------------------------------------------------------------------------------ helpers
    def _check(self, rc, what):
        if rc != 0:
            raise TempestHipError(f"{what}: {self.lib.tphu_last_error().decode()}")

    @staticmethod
    def _stream(t):
        import torch
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def _soa(self, a):
        """(n, d) rows [or one (d,) row] -> (tensor (d, n) contiguous, was_numpy, was_1d).  Host inputs go to `self.device`
        (set by the sampler that owns this object; the current device otherwise)."""
        import torch
        was_np = not isinstance(a, torch.Tensor)
        dev = getattr(self, "device", None) or torch.device("cuda", torch.cuda.current_device())
        if was_np:
            a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        one = a.dim() == 1
        if one:
            a = a.reshape(1, -1)
        if a.dim() != 2 or a.shape[1] != self.n_dim:
            raise ValueError(f"expected (..., {self.n_dim}) points, got {tuple(a.shape)}")
        if not a.is_cuda:
            a = a.to(dev)
        if a.dtype != torch.float64:
            a = a.to(torch.float64)
        t = a.T
        return (t if t.is_contiguous() else t.contiguous()), was_np, one

    # ---------------------------------------------------------------------------------- callbacks
    def prior_transform(self, u):
        """(n, n_dim) unit-cube points [or one point] -> parameters, same container kind as the input."""
        import torch
        us, was_np, one = self._soa(u)
        n = us.shape[1]
        xs = torch.empty_like(us)
        self._check(self.lib.tphu_prior(self._stream(us), us.data_ptr(), n, n, xs.data_ptr(), n), "tphu_prior")
        x = xs.T                                   # (n, d) strided view of the SoA buffer: no copy on the way back
        x = x[0] if one else x
        return x.cpu().numpy() if was_np else x

    def log_likelihood(self, x):
        import torch
        xs, was_np, one = self._soa(x)
        n = xs.shape[1]
        ll = torch.empty(n, dtype=torch.float64, device=xs.device)
        self._check(self.lib.tphu_like(self._stream(xs), xs.data_ptr(), n, n, ll.data_ptr()), "tphu_like")
        ll = ll[0] if one else ll
        return ll.cpu().numpy() if was_np else ll

    # ------------------------------------------------------------------------------ fused MCMC step
    def accept(self, kernel_id, beta, u, x, logl, uprime, maha_u, maha_up, assign, K, dof, seed, tick, item0, sums,
               ctl=None, partials=None, pending=None):
        """tph_accept with the two callbacks evaluated inside the kernel (u, x: (d, n) SoA tensors, updated in place)."""
        n = u.shape[1]
        if partials is None or partials.numel() < ((n + 255) // 256) * (1 + K):
            raise TempestHipError("HipCallbacks.accept: partials buffer missing or too small")
        for t in (u, logl, uprime) + ((x,) if x is not None else ()):
            if not (t.is_cuda and t.is_contiguous()):
                raise TempestHipError("HipCallbacks.accept: expected contiguous device tensors")
        p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        self._check(self.lib.tphu_accept(self._stream(u), int(kernel_id), float(beta), p(u), p(x), p(logl), p(uprime),
                                         p(maha_u), p(maha_up), p(assign), n, n, int(K), p(dof), int(seed), int(tick),
                                         int(item0), p(sums), p(ctl), p(partials), p(pending)), "tphu_accept")


    def can_fuse_step(self, K, has_assign, n) -> bool:
        """The whole step (proposal + callbacks + Metropolis update) in ONE kernel: register proposal kernel only
        (n_dim <= 16), one proposal mode, and shards up to 512 K particles -- measured: 45 -> 41 us per step at 131 072
        particles, where the step is latency-bound, but 157 -> 165 us at 1 048 576, where the proposal kernel is VALU-bound
        and the longer kernel only lowers its occupancy."""
        return (self.fused and self.whole_step and self.n_dim <= 16 and K == 1 and not has_assign
                and (self.whole_step == "always" or n <= 512 * 1024))

    def can_run(self, K, has_assign, n) -> bool:
        """A whole run of steps (the loop of mcmc.py:142-208) in ONE cooperative launch: where the whole-step kernel applies
        (one process; the caller checks that), unless a launch was refused before (no cooperative launch on the device)."""
        return bool(self.persistent) and self.can_fuse_step(K, has_assign, n)

    def run(self, kernel_id, u, logl, maha_u, modes, sigmas, bc, seed, tick_propose, tick_accept, item0, ctl, partials2, barrier,
            counts, n_global, n_steps, n_max, mailbox, slots, max_steps, redraw_lanes=0) -> bool:
        """tphu_run: steps until the stopping rule of `ctl` fires, adaptation included, in one launch (partials2: 2 x tiles x 2
        doubles; barrier: 4 int32 words; mailbox: (slots + 1) x 8 pinned doubles, the last row takes the final record).  False:
        the device refused the launch -- nothing ran, step the usual way."""
        n = u.shape[1]
        tiles = (n + 255) // 256
        if partials2.numel() < 4 * tiles or barrier.numel() < 4 or mailbox.numel() < 8 * (slots + 1):
            raise TempestHipError("HipCallbacks.run: buffers too small")
        for t in (u, logl, maha_u):
            if not (t.is_cuda and t.is_contiguous()):
                raise TempestHipError("HipCallbacks.run: expected contiguous device tensors")
        p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        winv = getattr(modes, "winv_dev", None)
        if winv is None:
            import torch
            winv = torch.linalg.inv(modes.chol_dev)
        rc = self.lib.tphu_run(self._stream(u), int(kernel_id), p(u), p(logl), p(maha_u), n, n, p(modes.means_dev), p(modes.chol_dev),
                               p(winv), p(modes.dof_dev), p(sigmas), p(bc), int(seed), int(tick_propose), int(tick_accept), int(item0),
                               p(ctl), p(partials2), p(barrier), p(counts), float(n_global), int(n_steps), int(n_max), p(mailbox),
                               int(slots), int(max_steps), int(redraw_lanes), int(self.run_groups))
        if rc == -3:
            self.persistent = False
            return False
        self._check(rc, "tphu_run")
        return True

    def step(self, kernel_id, u, logl, maha_u, modes, sigmas, bc, seed, tick_propose, tick_accept, item0, ctl, partials,
             redraw_lanes=0):
        """tph_propose + tphu_accept in one launch (u: (d, n) SoA, updated in place; x is not maintained)."""
        n = u.shape[1]
        if partials is None or partials.numel() < ((n + 255) // 256) * 2:
            raise TempestHipError("HipCallbacks.step: partials buffer missing or too small")
        for t in (u, logl, maha_u):
            if not (t.is_cuda and t.is_contiguous()):
                raise TempestHipError("HipCallbacks.step: expected contiguous device tensors")
        p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        winv = getattr(modes, "winv_dev", None)
        if winv is None:               # mode statistics built outside ModeStatistics: L^-1 from the factors
            import torch
            winv = torch.linalg.inv(modes.chol_dev)
        self._check(self.lib.tphu_step(self._stream(u), int(kernel_id), 0.0, p(u), p(logl), p(maha_u), n, n,
                                       p(modes.means_dev), p(modes.chol_dev), p(winv), p(modes.dof_dev), p(sigmas),
                                       p(bc), int(seed), int(tick_propose), int(tick_accept), int(item0), p(ctl), p(partials),
                                       int(redraw_lanes)), "tphu_step")


def fused_plugin(prior_transform, log_likelihood):
    """The HipCallbacks object both callbacks belong to (and that allows fusion), else None."""
    a = getattr(prior_transform, "__self__", None)
    b = getattr(log_likelihood, "__self__", None)
    if isinstance(a, HipCallbacks) and a is b and a.fused:
        return a
    return None

"""Thin Python handle on one `tph_ctx` (one per GPU): every numeric step of the hot path goes
through libtempest_hip.so here.  torch is only the device-array container (allocation, the current
HIP stream, the tensors handed to the user's callbacks); no torch op computes anything below.

Layout convention: particle arrays are SoA / dimension-major tensors of shape (n_dim, n) (row j =
coordinate j of every particle), FP64; index arrays int64; labels int32.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

KERNEL_ID = {"tpcn": 0, "rwm": 1}
KEY_U, KEY_X, KEY_LOGL, KEY_LOGMIX = 0, 1, 2, 3
STEP_STATE_LEN = 10            # TPH_STEP_STATE_LEN
OPT_ML_UNSTAGED = 3            # TPH_OPT_ML_UNSTAGED
OPT_BLOCKED = 4                # TPH_OPT_BLOCKED
OPT_MODES_EPOCH = 5            # TPH_OPT_MODES_EPOCH
OPT_ROW_MIRROR = 6             # TPH_OPT_ROW_MIRROR
OPT_COV_KERNEL = 7             # TPH_OPT_COV_KERNEL
OPT_SORTED_DRAWS = 8           # TPH_OPT_SORTED_DRAWS
OPT_STAGED_REDRAW = 9          # TPH_OPT_STAGED_REDRAW
OPT_SM_LANES = 10              # TPH_OPT_SM_LANES
OPT_SM_THRESHOLD = 11          # TPH_OPT_SM_THRESHOLD
OPT_SCREEN = 12                # TPH_OPT_SCREEN
OPT_MF_LANES = 13              # TPH_OPT_MF_LANES
OPT_MF_AUDIT = 14              # TPH_OPT_MF_AUDIT
OPT_BLK_MFMA = 15              # TPH_OPT_BLK_MFMA
OPT_BLK_TRIES = 16             # TPH_OPT_BLK_TRIES
OPT_BLK_FAN = 17               # TPH_OPT_BLK_FAN
OPT_HISTORY_VM = 18            # TPH_OPT_HISTORY_VM
OPT_MF_DEAL = 19               # TPH_OPT_MF_DEAL
OPT_BLK_STAGE = 20             # TPH_OPT_BLK_STAGE
OPT_GMM_KERNEL = 21            # TPH_OPT_GMM_KERNEL
OPT_FORMS_MFMA = 22            # TPH_OPT_FORMS_MFMA
BC_STRICT, BC_PERIODIC, BC_REFLECTIVE = 0, 1, 2

TAG_PRIOR, TAG_NORMAL, TAG_GAMMA, TAG_ACCEPT, TAG_RESAMPLE, TAG_UPSAMPLE, TAG_REPAIR, TAG_SYST = 1, 2, 3, 4, 5, 6, 7, 8


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.TempestHipError("expected a device tensor")
    if dtype is not None and t.dtype != dtype:
        raise _lib.TempestHipError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.TempestHipError("expected a contiguous tensor")
    return t.data_ptr()


def vshards_for(n_global: int) -> int:
    """Virtual shards of the canonical partition (csrc/common.h: tph_vshards_for)."""
    for v in (48, 16, 12, 8, 6, 4, 3, 2, 1):
        if n_global > 0 and n_global % (v * 256) == 0:
            return v
    return 1


def _hptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HipContext:
    """Owns the on-device persistent ensemble (history) for one GPU."""

    def __init__(self, n_dim, device=None, capacity_hint=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.TempestHipError("no GPU visible: tempest_amd has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.n_dim = int(n_dim)
        torch.cuda.set_device(self.device)
        self._ctx = C.c_void_p()
        self.rows_hint = int(capacity_hint)         # rows the history is reserved for (0: grows on demand)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.tph_ctx_create(self.device.index, self.n_dim, int(capacity_hint), C.c_void_p(stream),
                                      C.byref(self._ctx)), "tph_ctx_create")

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.tph_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ allocation helpers
    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def empty_rows(self, n, dtype=torch.float64):
        """1-D scratch of `n` entries for arrays that grow with the history (weights, cdf, multiplicities): the backing
        block is rounded up to 2^k or 1.5 * 2^k entries, so torch's caching allocator hands the same block back while the
        history grows inside a bucket instead of calling hipMalloc for a slightly larger one every iteration."""
        n = int(n)
        p2 = 1 << max(10, (n - 1).bit_length())          # next power of two >= n
        cap = p2 * 3 // 4 if p2 * 3 // 4 >= n else p2
        # with a reserved history (rows_hint) every such array is sized for THAT from the start: a handful of blocks are
        # allocated once, at the first iteration that needs them, instead of a new bucket every time the history doubles
        # (each a hipMalloc of up to half a gigabyte in the middle of the run)
        if n <= self.rows_hint:
            cap = max(cap, self.rows_hint)
        return torch.empty(cap, dtype=dtype, device=self.device)[:n]

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.tph_set_stream(self._ctx, C.c_void_p(s)), "tph_set_stream")

    def set_option(self, option, value):
        check(self.lib.tph_set_option(self._ctx, int(option), int(value)), "tph_set_option")

    def warmup(self):
        """Load every code object of the library now (tph_warmup) instead of at each kernel family's first use."""
        check(self.lib.tph_warmup(self._ctx), "tph_warmup")

    def synchronize(self):
        check(self.lib.tph_synchronize(self._ctx), "tph_synchronize")

    # ------------------------------------------------------------------------------ history
    @property
    def size(self):
        return int(self.lib.tph_history_size(self._ctx))

    @property
    def iterations(self):
        return int(self.lib.tph_history_iterations(self._ctx))

    def history_append(self, u, x, logl, beta, logz, n_global=None):
        n = logl.shape[0]
        ld = u.shape[1]
        check(self.lib.tph_history_append(self._ctx, _ptr(u, torch.float64), _ptr(x, torch.float64),
                                          _ptr(logl, torch.float64), n, ld, float(beta), float(logz),
                                          int(n if n_global is None else n_global)), "tph_history_append")

    def history_ptr(self, key):
        """(device pointer, leading dimension) of a history array."""
        p, ld = C.c_void_p(), C.c_int64()
        check(self.lib.tph_history_ptr(self._ctx, key, C.byref(p), C.byref(ld)), "tph_history_ptr")
        return p.value, ld.value

    def last_error(self) -> str:
        msg = self.lib.tph_last_error()
        return msg.decode() if msg else ""

    def history_memory(self):
        """Where the history's memory is (tph_history_memory): a dict of row counts and growth counters."""
        out = np.zeros(9, dtype=np.int64)
        check(self.lib.tph_history_memory(self._ctx, _hptr(out)), "tph_history_memory")
        keys = ("rows", "rows_backed", "rows_reserved", "mapped", "mirror_rows", "growth_steps", "rereservations", "mirror_drops", "copies")
        return dict(zip(keys, (int(v) for v in out)))

    def history_clear(self):
        self._vv_centre = None
        check(self.lib.tph_history_clear(self._ctx), "tph_history_clear")

    def history_load(self, u, x, logl, beta_t, logz_t, n_t, n_t_global=None, soa=False):
        """u, x: host arrays (N_h, d) -- or (d, N_h), the device layout, with soa=True -- or None; logl (N_h,)."""
        logl = np.ascontiguousarray(logl, dtype=np.float64)
        n = logl.size
        self._vv_centre = None
        tr = (lambda a: a) if soa else (lambda a: a.T)
        ut = np.ascontiguousarray(tr(np.asarray(u, dtype=np.float64))) if u is not None else None
        xt = np.ascontiguousarray(tr(np.asarray(x, dtype=np.float64))) if x is not None else None
        for a in (ut, xt):
            if a is not None and a.shape != (self.n_dim, n):
                raise _lib.TempestHipError(f"history_load: array of shape {a.shape}, expected {(self.n_dim, n)}")
        bt = np.ascontiguousarray(beta_t, dtype=np.float64)
        zt = np.ascontiguousarray(logz_t, dtype=np.float64)
        nt = np.ascontiguousarray(n_t, dtype=np.int64)
        ng = np.ascontiguousarray(n_t if n_t_global is None else n_t_global, dtype=np.int64)
        check(self.lib.tph_history_load(self._ctx, _hptr(ut) if ut is not None else None,
                                        _hptr(xt) if xt is not None else None, _hptr(logl), n, bt.size,
                                        _hptr(bt), _hptr(zt), _hptr(nt), _hptr(ng)), "tph_history_load")

    def history_read(self, key, off=0, n=None, soa=False):
        """Host copy: (n, d) C-contiguous for u/x -- (d, n), the device layout, with soa=True --, (n,) for logl/logmix."""
        n = self.size - off if n is None else n
        if key in (KEY_U, KEY_X):
            buf = np.empty((self.n_dim, n), dtype=np.float64)
        else:
            buf = np.empty((n,), dtype=np.float64)
        if n > 0:
            check(self.lib.tph_history_read(self._ctx, key, off, n, _hptr(buf)), "tph_history_read")
        return np.ascontiguousarray(buf.T) if (buf.ndim == 2 and not soa) else buf

    # --------------------------------------------------------------------------- reweighting
    def reweight_eval(self, betas):
        """[(vmax, s1, s2)] per trial beta (host floats; synchronises)."""
        b = np.ascontiguousarray(np.atleast_1d(betas), dtype=np.float64)
        out = np.empty((b.size, 3), dtype=np.float64)
        check(self.lib.tph_reweight_eval(self._ctx, _hptr(b), b.size, _hptr(out)), "tph_reweight_eval")
        return out

    def reweight_partials(self, betas, out=None):
        b = np.ascontiguousarray(np.atleast_1d(betas), dtype=np.float64)
        if out is None:
            out = self.empty(b.size, 3)
        check(self.lib.tph_reweight_partials(self._ctx, _hptr(b), b.size, _ptr(out, torch.float64)),
              "tph_reweight_partials")
        return out

    def reweight_time(self, beta=0.37, nb=1, reps=20):
        """Average launch duration (ms) of the reduction kernel, HIP events on the ctx stream."""
        out = C.c_double(0.0)
        check(self.lib.tph_bench_reweight_time(self._ctx, float(beta), int(nb), int(reps), C.byref(out)), "tph_bench_reweight_time")
        return out.value

    # ------------------------------------------------------------------ small collectives inside the library
    P2P_HANDLE_BYTES = 64

    def p2p_export(self) -> bytes:
        """This rank's inbox for the peer-to-peer small-message collectives (tph_comm_p2p_export): a HIP IPC handle."""
        h = C.create_string_buffer(self.P2P_HANDLE_BYTES)
        check(self.lib.tph_comm_p2p_export(self._ctx, h), "tph_comm_p2p_export")
        return h.raw

    def p2p_attach(self, handles) -> bool:
        """Map the peers' inboxes (handles in rank order) and self-test; True on every rank or False on every rank."""
        blob = b"".join(handles)
        ok = C.c_int(0)
        check(self.lib.tph_comm_p2p_attach(self._ctx, C.c_char_p(blob), C.byref(ok)), "tph_comm_p2p_attach")
        return bool(ok.value)

    @property
    def p2p_active(self) -> bool:
        return bool(self.lib.tph_comm_p2p_active(self._ctx))

    def comm_stats(self, reset=False):
        """Traffic counters of the ctx (tph_comm_stats): dict of p2p exchanges, callback collectives / bytes, shuffled rows / bytes."""
        out = (C.c_int64 * 5)()
        check(self.lib.tph_comm_stats(self._ctx, out, 1 if reset else 0), "tph_comm_stats")
        return dict(zip(("p2p_exchanges", "callback_collectives", "callback_bytes", "shuffle_rows", "shuffle_bytes"), [int(v) for v in out]))

    def p2p_status(self):
        check(self.lib.tph_comm_p2p_status(self._ctx), "tph_comm_p2p_status")

    def allreduce_dev(self, t, op=0):
        """In-place all-reduce (0 sum, 1 max, 2 min) of a small contiguous device tensor over the attached communicator,
        on the ctx stream (tph_comm_allreduce_dev)."""
        dt = {torch.float64: 0, torch.int64: 1, torch.int32: 2}[t.dtype]
        assert t.is_contiguous() and t.is_cuda
        check(self.lib.tph_comm_allreduce_dev(self._ctx, C.c_void_p(t.data_ptr()), t.numel(), dt, int(op)),
              "tph_comm_allreduce_dev")
        return t

    def resample_put_global(self, idx, n_local, u=None, x=None, logl=None):
        """One-sided resample shuffle (tph_resample_put_global): idx from resample_select_global over all n_local * world
        slots; returns this rank's (u, x, logl) in slot order, or None when the ranks agreed that the windows cannot be mapped."""
        d, n_local = self.n_dim, int(n_local)
        if u is None:
            u, x, logl = self.empty(d, n_local), self.empty(d, n_local), self.empty(n_local)
        rc = self.lib.tph_resample_put_global(self._ctx, _ptr(idx, torch.int64), idx.numel(), n_local, _ptr(u), _ptr(x),
                                              _ptr(logl), u.shape[1])
        if rc == 1:            # agreed between the ranks: no row windows on this node -> the caller's all-to-all
            return None
        check(rc, "tph_resample_put_global")
        return u, x, logl

    def membw_time(self, mode, n_doubles, reps=20):
        """Average launch duration (ms) of the streaming read (mode 0) / copy (mode 1) ceiling kernel."""
        out = C.c_double(0.0)
        check(self.lib.tph_bench_membw_time(self._ctx, int(mode), int(n_doubles), int(reps), C.byref(out)), "tph_bench_membw_time")
        return out.value

    def fp64_tflops(self, reps=10):
        out = C.c_double(0.0)
        check(self.lib.tph_bench_fp64_time(self._ctx, int(reps), C.byref(out)), "tph_bench_fp64_time")
        return out.value

    def mf_normals_error(self, seed, first, n_blocks, edge=False):
        """(max |z~ - z|, max |z|, blocks) of the screened kernel's FP32 Box-Muller pair against the FP64 pair."""
        out = (C.c_double * 3)()
        check(self.lib.tph_bench_mf_normals(self._ctx, int(seed), int(first), int(n_blocks), int(bool(edge)), out), "tph_bench_mf_normals")
        return out[0], out[1], out[2]

    def mf_counters(self):
        """Counters of the last screened proposal launch: dict(attempts, particles, contradictions, verified, screened, pair_jobs)."""
        out = (C.c_ulonglong * 7)()
        check(self.lib.tph_bench_mf_counters(self._ctx, out), "tph_bench_mf_counters")
        return dict(attempts=out[1], particles=out[2], contradictions=out[3], verified=out[4], screened=out[5], pair_jobs=out[6])

    def weights(self, beta, vmax, s1, out=None):
        if out is None:
            out = self.empty_rows(self.size)
        check(self.lib.tph_weights(self._ctx, float(beta), float(vmax), float(s1), _ptr(out, torch.float64)),
              "tph_weights")
        return out

    def logw(self, beta, n_h_global=None, out=None):
        if out is None:
            out = self.empty_rows(self.size)
        check(self.lib.tph_logw(self._ctx, float(beta), int(self.size if n_h_global is None else n_h_global),
                                _ptr(out, torch.float64)), "tph_logw")
        return out

    def sum_sq_max(self, w):
        out = np.empty(3)
        check(self.lib.tph_sum_sq_max(self._ctx, _ptr(w, torch.float64), w.numel(), _hptr(out)), "tph_sum_sq_max")
        return out

    # ------------------------------------------------------------------------------- trimming
    def trim_threshold(self, w, ess=0.99, bins=1000, sync=False, global_=False):
        """Device tensor (threshold, kept_sum, kept_count, ess_total) [+ host copy if sync].  global_: the threshold of
        the GLOBAL weight vector of a sharded history (the plain function without a communicator)."""
        out = self.empty(4)
        host = np.empty(4) if sync else None
        fn = self.lib.tph_trim_threshold_global if global_ else self.lib.tph_trim_threshold
        check(fn(self._ctx, _ptr(w, torch.float64), w.numel(), float(ess), int(bins), _ptr(out),
                 _hptr(host) if sync else None), "tph_trim_threshold")
        return (out, host) if sync else out

    # ------------------------------------------------- global order over a sharded history (tph_comm_attach)
    def cdf_global(self, w, thr=None, out=None, total=False):
        """This rank's slice of the global cumulative weight (the plain cdf without a communicator) [, global total]."""
        if out is None:
            out = self.empty_rows(w.numel())
        tot = C.c_double(0.0)
        check(self.lib.tph_cdf_global(self._ctx, _ptr(w, torch.float64), w.numel(), _ptr(thr), _ptr(out),
                                      C.byref(tot) if total else None), "tph_cdf_global")
        return (out, tot.value) if total else out

    def resample_select_global(self, cdf, n_slots, scheme, seed, tick, u0=0.0, pscale=1.0, tag=TAG_RESAMPLE):
        idx = self.empty(n_slots, dtype=torch.int64)
        check(self.lib.tph_resample_select_global(self._ctx, _ptr(cdf), cdf.numel(), n_slots, int(scheme), seed, tick, tag,
                                                  float(u0), float(pscale), _ptr(idx)), "tph_resample_select_global")
        return idx

    def multinomial_counts_global(self, cdf, seed, tick, kept_count=None, factor=4, n_draw_max=None, tag=TAG_UPSAMPLE):
        n = cdf.numel()
        counts = self.empty_rows(n, dtype=torch.int32)
        if n_draw_max is None:
            n_draw_max = factor * n
        check(self.lib.tph_multinomial_counts_global(self._ctx, _ptr(cdf), n, _ptr(kept_count), factor, n_draw_max, seed,
                                                     tick, tag, _ptr(counts)), "tph_multinomial_counts_global")
        return counts

    # ----------------------------------------------------------------------------- resampling
    def cdf(self, w, thr=None, out=None):
        if out is None:
            out = self.empty_rows(w.numel())
        check(self.lib.tph_cdf(self._ctx, _ptr(w, torch.float64), w.numel(), _ptr(thr), _ptr(out)), "tph_cdf")
        return out

    def resample_systematic(self, cdf, n_out, u0, i0=0, size_global=None, renorm=1.0):
        idx = self.empty(n_out, dtype=torch.int64)
        check(self.lib.tph_resample_systematic(self._ctx, _ptr(cdf), cdf.numel(), n_out, i0,
                                               n_out if size_global is None else size_global, float(u0),
                                               float(renorm), _ptr(idx)), "tph_resample_systematic")
        return idx

    def resample_multinomial(self, cdf, n_out, seed, tick, tag=TAG_RESAMPLE, item0=0):
        idx = self.empty(n_out, dtype=torch.int64)
        check(self.lib.tph_resample_multinomial(self._ctx, _ptr(cdf), cdf.numel(), n_out, seed, tick, tag, item0,
                                                _ptr(idx)), "tph_resample_multinomial")
        return idx

    def resample_select(self, cdf, n_slots, scheme, seed, tick, u0, w_before, w_upto, w_total, is_last,
                        tag=TAG_RESAMPLE):
        idx = self.empty(n_slots, dtype=torch.int64)
        check(self.lib.tph_resample_select(self._ctx, _ptr(cdf), cdf.numel(), n_slots, int(scheme), seed, tick, tag,
                                           float(u0), float(w_before), float(w_upto), float(w_total), int(is_last),
                                           _ptr(idx)), "tph_resample_select")
        return idx

    def gather(self, idx, u_out, x_out, logl_out):
        check(self.lib.tph_gather(self._ctx, _ptr(idx, torch.int64), idx.numel(), _ptr(u_out), _ptr(x_out),
                                  _ptr(logl_out), u_out.shape[1]), "tph_gather")

    def posterior_rows(self, idx, m, w=None, wdiv=1.0, key=KEY_X):
        """Rows `idx` (None: the first m) of the history as device tensors x (m, d) row-major, logl (m,), w/wdiv (m,)."""
        x, logl = self.empty(m, self.n_dim), self.empty(m)
        wout = self.empty(m) if w is not None else None
        if m > 0:
            check(self.lib.tph_posterior_rows(self._ctx, key, _ptr(idx, torch.int64), m, _ptr(w, torch.float64), float(wdiv),
                                              _ptr(x), _ptr(logl), _ptr(wout)), "tph_posterior_rows")
        return x, logl, wout

    def index_compose(self, a, b):
        out = self.empty(b.numel(), dtype=torch.int64)
        check(self.lib.tph_index_compose(self._ctx, _ptr(a, torch.int64), _ptr(b, torch.int64), b.numel(), _ptr(out)),
              "tph_index_compose")
        return out

    def multinomial_counts(self, cdf, seed, tick, kept_count=None, factor=4, n_draw_max=None, tag=TAG_UPSAMPLE):
        n = cdf.numel()
        counts = self.empty_rows(n, dtype=torch.int32)
        if n_draw_max is None:
            n_draw_max = factor * n
        check(self.lib.tph_multinomial_counts(self._ctx, _ptr(cdf), n, _ptr(kept_count), factor, n_draw_max, seed,
                                              tick, tag, _ptr(counts)), "tph_multinomial_counts")
        return counts

    # ------------------------------------------------------------------------------- mutation
    def prior_draw(self, u, seed, tick, item0=0):
        check(self.lib.tph_prior_draw(self._ctx, _ptr(u, torch.float64), u.shape[1], u.shape[1], seed, tick, item0),
              "tph_prior_draw")

    def inf_repair(self, u, x, logl, seed, tick, item0=0, return_src=False):
        """(n_finite, n) on the device; with `return_src` also src (int64, n): the row that now sits in each row."""
        stats = self.empty(2)
        if not return_src:
            check(self.lib.tph_inf_repair(self._ctx, _ptr(u), _ptr(x), _ptr(logl), logl.numel(), u.shape[1], seed, tick,
                                          item0, _ptr(stats)), "tph_inf_repair")
            return stats
        src = torch.empty(logl.numel(), dtype=torch.int64, device=self.device)
        check(self.lib.tph_inf_repair_src(self._ctx, _ptr(u), _ptr(x), _ptr(logl), logl.numel(), u.shape[1], seed, tick,
                                          item0, _ptr(stats), _ptr(src, torch.int64)), "tph_inf_repair_src")
        return stats, src

    def propose(self, kernel, u, assign, modes, sigmas, bc, seed, tick, item0, uprime, maha_u, maha_up, ctl=None,
                pending=None):
        n = u.shape[1]
        if ctl is not None and ctl.numel() < STEP_STATE_LEN:
            raise _lib.TempestHipError(f"propose: the step-control block needs {STEP_STATE_LEN} doubles")
        check(self.lib.tph_propose(self._ctx, KERNEL_ID[kernel], _ptr(u), _ptr(assign, torch.int32) if assign is not None else None,
                                   n, n, modes.K, _ptr(modes.means_dev), _ptr(modes.chol_dev),
                                   _ptr(getattr(modes, "winv_dev", None)),
                                   _ptr(modes.dof_dev), _ptr(sigmas), _ptr(bc) if bc is not None else None, seed, tick,
                                   item0, _ptr(uprime), _ptr(maha_u), _ptr(maha_up),
                                   _ptr(ctl) if ctl is not None else None,
                                   _ptr(pending, torch.uint8) if pending is not None else None), "tph_propose")

    def accept(self, kernel, beta, u, x, logl, uprime, xprime, loglprime, maha_u, maha_up, assign, K, dof, seed,
               tick, item0, sums, ctl=None, partials=None, pending=None):
        n = u.shape[1]
        if partials is not None and partials.numel() < ((n + 255) // 256) * (1 + K):
            raise _lib.TempestHipError("accept: partials buffer too small")
        check(self.lib.tph_accept(self._ctx, KERNEL_ID[kernel], float(beta), _ptr(u), _ptr(x), _ptr(logl),
                                  _ptr(uprime), _ptr(xprime, torch.float64) if x is not None else None,
                                  _ptr(loglprime, torch.float64),
                                  _ptr(maha_u), _ptr(maha_up),
                                  _ptr(assign, torch.int32) if assign is not None else None, n, n, K, _ptr(dof), seed,
                                  tick, item0, _ptr(sums) if sums is not None else None,
                                  _ptr(ctl) if ctl is not None else None,
                                  _ptr(partials) if partials is not None else None,
                                  _ptr(pending, torch.uint8) if pending is not None else None), "tph_accept")

    def adapt(self, kernel, sums, counts, K, n_global, n_steps, n_max, sigmas, state, mailbox=None, partials=None, n=0):
        """mailbox: pinned host tensor (slots, 8) the step record is also written to (polled by the host);
        partials: accept()'s block partials of n particles, column-summed here (accept called with sums=None)."""
        if mailbox is not None and not (mailbox.is_pinned() and mailbox.dtype == torch.float64 and mailbox.is_contiguous()):
            raise _lib.TempestHipError("adapt: mailbox must be a pinned contiguous float64 host tensor")
        if mailbox is not None and state.numel() < STEP_STATE_LEN:
            raise _lib.TempestHipError(f"adapt: with a mailbox the state block needs {STEP_STATE_LEN} doubles")
        check(self.lib.tph_adapt(self._ctx, KERNEL_ID[kernel], _ptr(sums), _ptr(counts), K, float(n_global),
                                 self.n_dim, int(n_steps), int(n_max), _ptr(sigmas), _ptr(state),
                                 mailbox.data_ptr() if mailbox is not None else None,
                                 mailbox.shape[0] if mailbox is not None else 0,
                                 _ptr(partials) if partials is not None else None, int(n)), "tph_adapt")

    def accept_sums_global(self, partials, n, K, sums, host_paced=False):
        """The step's (#accepted, sum alpha_c) over all ranks from accept()'s block partials, in the canonical shard order
        (tph_accept_sums_global); the one rank's sums without a communicator."""
        check(self.lib.tph_accept_sums_global(self._ctx, _ptr(partials), int(n), int(K), _ptr(sums), 1 if host_paced else 0),
              "tph_accept_sums_global")
        return sums

    def cluster_counts(self, assign, n, K):
        out = self.empty(K)
        check(self.lib.tph_cluster_counts(self._ctx, _ptr(assign, torch.int32) if assign is not None else None, n, K,
                                          _ptr(out)), "tph_cluster_counts")
        return out

    # --------------------------------------------------------------------------- proposal fit
    def fit_modes(self, counts, labels=None, K=1, n=None, global_=False):
        """-> (means, covs, chol, inv, winv): winv = L^-1 per mode (the form the proposal kernels consume).
        global_: the fit of the GLOBAL up-sampled set of a sharded history (the plain fit without a communicator)."""
        d = self.n_dim
        n = self.size if n is None else n
        means, covs = self.empty(K, d), self.empty(K, d, d)
        chol, inv, winv = self.empty(K, d, d), self.empty(K, d, d), self.empty(K, d, d)
        fn = self.lib.tph_fit_modes_global if global_ else self.lib.tph_fit_modes
        check(fn(self._ctx, _ptr(counts, torch.int32),
                                     _ptr(labels, torch.int32) if labels is not None else None, n, K, _ptr(means),
                                     _ptr(covs), _ptr(chol), _ptr(inv), _ptr(winv)), "tph_fit_modes")
        return means, covs, chol, inv, winv

    def chol_inv(self, covs):
        """-> (chol, inv, winv) of K covariance matrices (covs is ridged in place where the factorisation fails)."""
        K = covs.shape[0]
        chol, inv, winv = torch.empty_like(covs), torch.empty_like(covs), torch.empty_like(covs)
        check(self.lib.tph_chol_inv(self._ctx, _ptr(covs), K, _ptr(chol), _ptr(inv), _ptr(winv)), "tph_chol_inv")
        return chol, inv, winv

    # ----------------------------------------------------------------------------- clustering
    def compact_indices(self, w, thr, m):
        idx = self.empty(m, dtype=torch.int64)
        check(self.lib.tph_compact_indices(self._ctx, _ptr(w), w.numel(), _ptr(thr), _ptr(idx)), "tph_compact_indices")
        return idx

    def gather_u_affine(self, idx, shift=None, scale=None, w=None):
        m = idx.numel()
        out = self.empty(self.n_dim, m)
        wout = self.empty(m)
        check(self.lib.tph_gather_u_affine(self._ctx, _ptr(idx, torch.int64), m, _ptr(shift), _ptr(scale), _ptr(w),
                                           _ptr(out), m, _ptr(wout)), "tph_gather_u_affine")
        return out, wout

    def affine(self, x, shift, scale):
        check(self.lib.tph_affine(self._ctx, _ptr(x), x.shape[1], x.shape[1], _ptr(shift), _ptr(scale)), "tph_affine")

    def x_weighted_sums(self, x, w, with_range=False):
        sums = self.empty(1 + self.n_dim)
        rng = self.empty(2 * self.n_dim) if with_range else None
        check(self.lib.tph_x_weighted_sums(self._ctx, _ptr(x), x.shape[1], x.shape[1], _ptr(w), _ptr(sums), _ptr(rng)),
              "tph_x_weighted_sums")
        return (sums, rng) if with_range else sums

    def x_weighted_cov(self, x, w, mean):
        cov = self.empty(self.n_dim * self.n_dim)
        check(self.lib.tph_x_weighted_cov(self._ctx, _ptr(x), x.shape[1], x.shape[1], _ptr(w), _ptr(mean), _ptr(cov)),
              "tph_x_weighted_cov")
        return cov

    def gmm_estep(self, x, sw, labels, label, params, K, mode, eps=1e-10, wr=None, label_out=None, stats=None,
                  shift=None, scale=None, n=None, ld=None):
        """x: (d, n) SoA tensor, or a raw device pointer (int) with explicit n and ld (e.g. the history)."""
        if isinstance(x, int):
            xp = x
        else:
            xp, n, ld = _ptr(x), x.shape[1], x.shape[1]
        check(self.lib.tph_gmm_estep(self._ctx, xp, ld, n, _ptr(sw), _ptr(labels, torch.int32) if labels is not None else None,
                                     int(label), int(K), _ptr(params), int(mode), float(eps), _ptr(shift), _ptr(scale), _ptr(wr),
                                     _ptr(label_out, torch.int32) if label_out is not None else None, _ptr(stats)),
              "tph_gmm_estep")

    def gmm_em_state(self, K):
        """A zeroed state block of the device-paced EM (tph_gmm_em_*) and the offsets of its parts."""
        d = self.n_dim
        total = int(self.lib.tph_gmm_em_state_doubles(d, int(K)))
        off = {"ctl": 0}
        o = 16 + K * (1 + d) + K * d + K * d * d + K * (2 + d + d * d)
        off["weights"], off["means"], off["covs"] = o, o + K, o + K + K * d
        assert off["covs"] + K * d * d == total
        return self.zeros(total), off

    def gmm_em_begin(self, x, K, wr, state):
        check(self.lib.tph_gmm_em_begin(self._ctx, _ptr(x), x.shape[1], x.shape[1], int(K), _ptr(wr), _ptr(state)), "tph_gmm_em_begin")

    def gmm_em_run(self, x, sw, labels, label, K, wr, state, reg, tol, max_iter, iters):
        check(self.lib.tph_gmm_em_run(self._ctx, _ptr(x), x.shape[1], x.shape[1], _ptr(sw),
                                      _ptr(labels, torch.int32) if labels is not None else None, int(label), int(K), _ptr(wr),
                                      _ptr(state), float(reg), float(tol), int(max_iter), int(iters)), "tph_gmm_em_run")

    # ------------------------------------------------------------------------ volume variation
    def weighted_moments(self, w):
        d = self.n_dim
        out = self.empty(d + d * d)
        check(self.lib.tph_weighted_moments(self._ctx, _ptr(w), w.numel(), _ptr(out)), "tph_weighted_moments")
        return out

    def weighted_sums(self, w):
        out = self.empty(1 + self.n_dim)
        check(self.lib.tph_weighted_sums(self._ctx, _ptr(w), w.numel(), _ptr(out)), "tph_weighted_sums")
        return out

    def weighted_cov_centered(self, w, mean):
        d = self.n_dim
        out = self.empty(d * d)
        check(self.lib.tph_weighted_cov_centered(self._ctx, _ptr(w), w.numel(), _ptr(mean.contiguous()), _ptr(out)),
              "tph_weighted_cov_centered")
        return out

    def weighted_moments_shifted(self, w, centre):
        """(sum w, mean[d], cov[d*d]) of the history's u in one pass, moments taken about `centre` (n_dim <= 12)."""
        d = self.n_dim
        out = self.empty(1 + d + d * d)
        check(self.lib.tph_weighted_moments_shifted(self._ctx, _ptr(w), w.numel(), _ptr(centre.contiguous()), _ptr(out)),
              "tph_weighted_moments_shifted")
        return out

    def volume_variation(self, w, centre=None):
        """tools.py:58-117 on the history under the normalised device weights w, d x d work included (one host wait)."""
        out = C.c_double(0.0)
        check(self.lib.tph_volume_variation(self._ctx, _ptr(w, torch.float64), w.numel(), _ptr(centre), C.byref(out)),
              "tph_volume_variation")
        return out.value

    def cv_sum(self, w, mean, covinv):
        out = self.empty(1)
        check(self.lib.tph_cv_sum(self._ctx, _ptr(w), w.numel(), _ptr(mean), _ptr(covinv), _ptr(out)), "tph_cv_sum")
        return out

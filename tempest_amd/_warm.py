"""One-time start-up work of a process, done in parallel behind the Sampler's construction.

The first PS iteration of a fresh process used to take 0.55 s on an MI355X where every later one takes milliseconds:
0.14 s for the first host-to-device copy, 0.10-0.13 s each for the code objects behind the first `pow`, the first
multiply / subtract, ... of the user's eager torch callbacks (the HIP runtime loads a code object when one of its kernels
is first launched), 0.09 s for creating the device context and reserving the history.  None of it depends on the other, and
loads issued from different threads overlap (measured: five first-use operations 0.41 s one after the other, 0.14 s from
five threads; torch's bindings and ctypes both release the GIL).  So `Sampler.__init__` starts this warm-up -- the copy
path, the families of elementwise / reduction kernels a vectorised likelihood is made of, the user's own callbacks on a
four-row dummy batch, the device context -- each in its own thread, and the first iteration waits for it
(`SamplerCore._ensure_callbacks`).  What it cost is kept in `Sampler.startup_breakdown` (bench.py reports it).

Nothing here computes anything the run uses: it is process initialisation moved off the critical path.  The callbacks see
one extra four-row call (as the backend probe of core.py already makes); TEMPEST_AMD_WARMUP=0 turns the whole thing off.
"""
import os
import threading
import time

_process_warm = False          # the code objects stay loaded for the life of the process: later Samplers skip the generic part
_lock = threading.Lock()


class Warmup:
    def __init__(self):
        self.threads, self.seconds, self.errors = [], {}, {}
        self.t_start = time.perf_counter()
        self.waited = 0.0
        self._joined = False

    def add(self, name, fn):
        def run():
            t0 = time.perf_counter()
            try:
                fn()
            except Exception as e:          # a warm-up never fails a run: the real call will raise where it belongs
                self.errors[name] = f"{type(e).__name__}: {e}"[:200]
            self.seconds[name] = round(time.perf_counter() - t0, 4)
        t = threading.Thread(target=run, name=f"tempest-amd-warm-{name}", daemon=True)
        self.threads.append(t)
        t.start()

    def wait(self):
        if self._joined:
            return
        t0 = time.perf_counter()
        for t in self.threads:
            t.join()
        self._joined = True
        self.waited = round(time.perf_counter() - t0, 4)
        self.wall = round(time.perf_counter() - self.t_start, 4)

    def breakdown(self):
        out = {"threads_s": dict(self.seconds), "main_thread_waited_s": self.waited,
               "wall_s_from_construction": getattr(self, "wall", None)}
        if self.errors:
            out["errors"] = dict(self.errors)
        return out


def start(config, state):
    """Kick off the warm-up for a Sampler under construction; returns a Warmup (possibly with nothing to wait for)."""
    global _process_warm
    w = Warmup()
    if os.environ.get("TEMPEST_AMD_WARMUP", "1") == "0":
        return w
    try:
        import torch
        if not torch.cuda.is_available():
            return w
    except Exception:
        return w
    import numpy as np
    dev = state.device
    d = config.n_dim
    with _lock:
        first = not _process_warm
        _process_warm = True
    if not first:
        return w

    def on_side_stream(fn):
        def run():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                fn()
                torch.cuda.current_stream(dev).synchronize()
        return run
    x = torch.empty(4, d, dtype=torch.float64, device=dev)     # contents irrelevant: only the launches matter
    # the copy engines and the pinned staging path
    w.add("copy_path", on_side_stream(lambda: torch.from_numpy(np.ones(8)).to(dev).cpu()))
    # the kernel families eager tensor code is made of (one code object each)
    # (only the handful nearly every vectorised likelihood launches: as the first process on a machine the loads are bound by
    # reading the code objects from disk, and every family the run does not need is bandwidth taken from those it does)
    for name, fn in (("binary_mul", lambda: x * x), ("binary_add_sub", lambda: (x + x) - x), ("pow", lambda: x ** 2),
                     ("reduce_sum", lambda: x.sum(dim=1)), ("fill", lambda: torch.zeros(4, dtype=torch.float64, device=dev))):
        w.add(name, on_side_stream(fn))
    # the user's own callbacks on a four-row batch (whatever else they launch); torch tensors in, like the run itself
    if config.backend in ("auto", "torch") and getattr(config, "vectorize", False):
        def callbacks():
            u = torch.linspace(0.05, 0.95, 4 * d, dtype=torch.float64, device=dev).reshape(d, 4).T
            xs = config.prior_transform(u)
            if isinstance(xs, torch.Tensor):
                config.log_likelihood(xs)
        w.add("user_callbacks", on_side_stream(callbacks))
    # the device context: library load, scratch, the history reservation (gigabytes of hipMalloc).  A sharded run attaches
    # its communicator inside that property -- collectives belong to the main thread, so only un-sharded runs do it here.
    if state.comm is None or not state.comm.active:
        def ctx():
            torch.cuda.set_device(dev)
            c = state.ctx
            # ... and the library's own code objects, ALL of them (one per translation unit: an empty launch of each loads it) --
            # met one by one in a run's first iterations they were its first reweight, fit, resample, d > 16 proposal, clustering fit
            c.warmup()
        w.add("device_context", ctx)
    # the kernel families the library's own host code launches through torch around a clustering fit / a several-modes fit
    # (cluster.py, modes.py: label bookkeeping): first used in the middle of iteration 4 they were 0.22 s of config 3's first run
    if getattr(config, "clustering", False):
        li = torch.arange(8, device=dev) % 3
        for name, fn in (("cl_bincount", lambda: torch.bincount(li, minlength=4).to(torch.float64)),
                         ("cl_cumsum", lambda: (torch.cumsum((li > 0).to(torch.int32), 0) - 1).clamp(min=0).to(torch.int32)),
                         ("cl_index", lambda: (li.to(torch.int32))[li.long()].contiguous()),
                         ("cl_compare_sum", lambda: int((li == 1).sum().item())),
                         ("cl_where", lambda: torch.where(x > 0, x, torch.zeros_like(x)).sum()),
                         ("cl_cat_stack", lambda: torch.cat([torch.stack([x.sum(), x.sum()]), x.reshape(-1)]).cpu()),
                         ("cl_softmax", lambda: torch.softmax(x, dim=0).T.contiguous()),
                         ("cl_nonzero", lambda: torch.nonzero(li >= 1).reshape(-1))):
            w.add(name, on_side_stream(fn))
    return w

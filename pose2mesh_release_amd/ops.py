"""Host-side launch layer: torch tensors in, C-ABI calls (include/p2m.h) out.

torch is plumbing here (device memory, streams); every arithmetic step of the hot path runs in
libp2m_hip.so.  There is no fallback: CPU tensors or a missing library raise.
"""
import ctypes
import os as _os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import P2MError, check

_vp = ctypes.c_void_p


def _p(t):
    return None if t is None else _vp(t.data_ptr())


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _req(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise P2MError(f"{name}: expected a contiguous fp32 tensor on the GPU (got "
                       f"{type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}); "
                       f"this path has no CPU implementation")
    return t


class KernelTimer:
    """Optional live per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
    Kernels are launched on torch's current stream, so torch.cuda.Event brackets exactly the launch.
    `work` is a number (FLOPs or bytes the launch has to move) or a pair (that, SURVEY.md 8(d)'s algorithmic figure
    for the same stage when the two differ - the basis stage after the fake-vertex split).
    Usage: ops.TIMER = KernelTimer(); ...; ops.TIMER.summary()."""

    def __init__(self):
        self.records = {}     # name -> list of (start, end, work)

    def bracket(self, name, work):
        return _Bracket(self, name, work)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            out[name] = {"launches": len(recs), "ms": ms,
                         "work": float(sum(w[0] if isinstance(w, tuple) else w for _, _, w in recs)),
                         "work_alg": float(sum(w[1] if isinstance(w, tuple) else w for _, _, w in recs)),
                         # algorithmic HBM bytes of the launches (third element of the work tuple), for `traffic`
                         "bytes": float(sum(w[2] if isinstance(w, tuple) and len(w) > 2 else 0.0 for _, _, w in recs))}
        return out


class _Bracket:
    def __init__(self, timer, name, work):
        self.t, self.name, self.work = timer, name, work

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()

    def __exit__(self, *a):
        self.e.record()
        self.t.records.setdefault(self.name, []).append((self.s, self.e, self.work))


class _NoBracket:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOB = _NoBracket()
TIMER = None
_tls = threading.local()      # .phase: "" in forward, "_bwd" in backward (backward launches share the GPU with the
                              # side-stream dW GEMMs); thread-local because nn.DataParallel runs one thread per replica


def set_phase(phase):
    _tls.phase = phase


def _timed(name, work):
    return _NOB if TIMER is None else TIMER.bracket(name + getattr(_tls, "phase", ""), work)


# Kernel-variant knobs (environment, read once).  The defaults are the measured-fastest forms;
# the other settings exist as the INDEPENDENT kernel set of the parity tests and as A/B forms of the algebraic shortcuts.
# Variants measured and removed in round 3 (results in DESIGN.md section 6): the 4-wave / 3-chunk-ring contractions, the
# BatchNorm-backward reduction in the contraction epilogue, the project-then-combine forward of un-pooled convs, operand
# prefetch on a helper stream, the gather-in-GEMM f32 kernel.
SPLIT_FAKE = _os.environ.get("P2M_SPLIT_FAKE", "1") == "1"
# Weight-gradient contractions on a side stream (nothing downstream in the backward reads them); P2M_DW_STREAM=main runs them
# on the main stream, serially (the A/B form; bench.py's `exclusive_serial` figures).  Measured in round 4 and removed again
# (DESIGN.md section 6): a LOW-priority helper stream (38.40 vs 38.22 ms), helper streams masked to 64 / 32 CUs (43.9 /
# 43.4 ms), and holding the big levels' jobs back until the backward reaches the coarse levels (38.5-38.8 vs 38.0 ms: the
# main queue is never idle, so the HBM-bound work only moves); serial: 39.2 ms.
DW_STREAM = _os.environ.get("P2M_DW_STREAM", "side")
if DW_STREAM not in ("side", "main"):
    raise ValueError(f"P2M_DW_STREAM must be side or main, not {DW_STREAM!r}")
DW_SIDE_STREAM = DW_STREAM != "main"
# backward of un-pooled convs at the coarse resolution (paired operator, include/p2m.h); 0 = at the fine resolution
# with a pair-sum afterwards (the A/B form, also the independent path of the B=256 parity test)
PAIR_BWD = _os.environ.get("P2M_PAIR_BWD", "1") == "1"
# classes of identical fake rows inside the coarse-to-fine stack (include/p2m.h): only one representative of every run of
# identical padding rows is computed; 0 = every row (the A/B form)
CLASSES = _os.environ.get("P2M_CLASSES", "1") == "1"
# Chebyshev basis inside the contraction (p2m_cheb_tile_gemm, include/p2m.h): the real rows of a conv on a level with a
# tile plan in ONE kernel, no T1 / T2 planes in HBM.  "auto" (default): where it measured faster than basis kernel + plane
# contraction on MI355X - forward-form launches (no planes written out) of the level's own / un-pooled plans on the big
# levels; "1": wherever supported (the A/B form of the parity tests); "0": never.  DESIGN.md section 6 has the numbers:
# 157 KB of LDS per block leave one block per CU, so the kernel only wins where the plane traffic it removes is large.
# BatchNorm + ReLU applied WHERE THE CONSUMER LOADS (include/p2m.h "activation on load"): between the two convs of a block on
# the split levels, in train mode, the activated tensor x = relu(bn(y)) is never written - the second conv's kernels (tile
# kernel, or LDS-staged basis kernel + plane 0 of the contraction; the fake-row contraction; the narrow final conv's
# projection) and its weight gradients read the raw y and apply fma + max on the way into LDS.  0 = the separate pass (A/B form).
FOLD_ACT = _os.environ.get("P2M_FOLD_ACT", "1") == "1"
TILE_GEMM = _os.environ.get("P2M_TILE_GEMM", "auto")
# Round 6: the BatchNorm-backward reduction of a block's first conv summed by the kernel that produces its incoming
# gradient (the dX tile kernel of the block's second conv) instead of a stand-alone pass over g and y; 0: the separate pass
# (opt-in: same-box A/B at B = 256 - 43.44 / 43.39 ms fused vs 43.33 / 43.37 separate: the planes-out tile kernels grow by
# what the stand-alone reduce passes cost, 0.73 vs ~0.8 ms)
BN_FUSE = _os.environ.get("P2M_BN_FUSE", "0") == "1"
if TILE_GEMM not in ("auto", "0", "1"):
    raise ValueError(f"P2M_TILE_GEMM must be auto, 0 or 1, not {TILE_GEMM!r}")
TILE_GEMM_MG_MIN_ROWS = 300    # "auto", f16x2 and N <= 128: every split level with a 128- or 64-wide output (SMPL-like: 6890,
                               # 3638, 1923 real rows; MANO-like: 778, 389).  1500 -> 300: MANO B=512 +1.4 %, SMPL +0.3 %
TILE_GEMM_MIN_ROWS = 3000      # "auto", otherwise: the two finest levels, forward-form launches only ...
TILE_GEMM_MIN_BATCH = 128      # ... and, for the VALU-gather kernel, enough sample groups per tile to amortise a block's tables
                               # (B = 64 inference: 4.14 vs 3.83 ms per batch with that kernel on, measured)


class DeviceGraph:
    """One coarsening level baked on one GPU: merged CSR of L and 2LL-I (p2m_graph_create).
    Replaces the torch sparse COO tensor + per-forward .cuda() of lib/models/meshnet.py:61-62,81."""

    def __init__(self, L, device):
        import scipy.sparse as sp
        if isinstance(L, torch.Tensor):  # torch sparse COO/CSR as the reference keeps after __init__
            Lc = L.coalesce() if L.layout == torch.sparse_coo else L.to_sparse_coo().coalesce()
            idx = Lc.indices().cpu().numpy()
            L = sp.coo_matrix((Lc.values().cpu().numpy(), (idx[0], idx[1])), shape=tuple(Lc.shape))
        csr = sp.csr_matrix(L, dtype=np.float64)
        csr.sum_duplicates()
        csr.sort_indices()
        if csr.shape[0] != csr.shape[1]:
            raise P2MError("graph Laplacian must be square")
        self.V = int(csr.shape[0])
        self.device = torch.device(device)
        rp = np.ascontiguousarray(csr.indptr, dtype=np.int32)
        ci = np.ascontiguousarray(csr.indices, dtype=np.int32)
        va = np.ascontiguousarray(csr.data.astype(np.float32))   # graph_utils.py:104: fp32 on device
        h = _vp()
        with torch.cuda.device(self.device):
            check(_lib.hip().p2m_graph_create(rp.ctypes.data_as(_vp), ci.ctypes.data_as(_vp), va.ctypes.data_as(_vp),
                                              self.V, int(ci.size), ctypes.byref(h)), "p2m_graph_create")
        self.handle = h
        info = (ctypes.c_int32 * 4)()
        check(_lib.hip().p2m_graph_info(self.handle, ctypes.byref(info)), "p2m_graph_info")
        self.nnz_L, self.nnz_merged, self.max_row = int(info[1]), int(info[2]), int(info[3])
        cnt, coef = (ctypes.c_int32 * 2)(), (ctypes.c_float * 2)()
        check(_lib.hip().p2m_graph_split_info(self.handle, ctypes.byref(cnt), ctypes.byref(coef)), "p2m_graph_split_info")
        self.n_real, self.n_fake = int(cnt[0]), int(cnt[1])
        self.fake_a, self.fake_b = float(coef[0]), float(coef[1])
        # split the launches into real / fake vertices when it pays (big levels with many padding vertices)
        self.split = SPLIT_FAKE and self.V >= 512 and self.n_fake >= 0.15 * self.V
        # paired operator (backward of un-pooled convs at the coarse resolution): row sets 3 / 4 over V/2 rows
        check(_lib.hip().p2m_graph_pair_info(self.handle, ctypes.byref(cnt)), "p2m_graph_pair_info")
        self.n_pair_real, self.n_pair_fake = int(cnt[0]), int(cnt[1])
        self.pair = PAIR_BWD and self.split and self.n_pair_real > 0
        nt = (ctypes.c_int32 * 3)()
        check(_lib.hip().p2m_graph_plan_info(self.handle, ctypes.byref(nt)), "p2m_graph_plan_info")
        self.plan_tiles = tuple(int(v) for v in nt)
        # binades of headroom the two-fp16-slice contractions give the Chebyshev planes of this level (include/p2m.h)
        self.plane_bits = int(_lib.hip().p2m_graph_plane_bits(self.handle, 0))

        self.classes = False
        self.n_fake_all = self.n_fake

    def set_size(self, row_set):
        return (self.n_real, self.n_fake, self.n_pair_real, self.n_pair_fake)[row_set - 1]

    def fake_ids_host(self):
        out = np.zeros(max(self.n_fake, 1), dtype=np.int32)
        check(_lib.hip().p2m_graph_fake_ids(self.handle, out.ctypes.data_as(_vp)), "p2m_graph_fake_ids")
        return out[:self.n_fake]

    def real_ids_host(self):
        """The real vertices in COMPACT ROW ORDER (row i of every compact plane belongs to vertex out[i]): a locality
        order since round 5, not the ascending one (include/p2m.h p2m_graph_real_ids)."""
        out = np.zeros(max(self.n_real, 1), dtype=np.int32)
        check(_lib.hip().p2m_graph_real_ids(self.handle, out.ctypes.data_as(_vp)), "p2m_graph_real_ids")
        return out[:self.n_real]

    def set_classes(self, rep_of):
        """Declare the runs of identical fake rows (include/p2m.h "classes"): afterwards row sets 2 / 4 hold the
        representatives only."""
        rep = np.ascontiguousarray(rep_of, dtype=np.int32)
        if rep.shape != (self.V,):
            raise P2MError("rep_of must have one entry per vertex")
        with torch.cuda.device(self.device):
            check(_lib.hip().p2m_graph_set_classes(self.handle, rep.ctypes.data_as(_vp)), "p2m_graph_set_classes")
        c3, c2 = (ctypes.c_int32 * 3)(), (ctypes.c_int32 * 2)()
        check(_lib.hip().p2m_graph_class_info(self.handle, ctypes.byref(c3)), "p2m_graph_class_info")
        self.classes, self.n_fake, self.n_fake_all = bool(c3[0]), int(c3[1]), int(c3[2])
        check(_lib.hip().p2m_graph_pair_info(self.handle, ctypes.byref(c2)), "p2m_graph_pair_info")
        self.n_pair_real, self.n_pair_fake = int(c2[0]), int(c2[1])

    def __del__(self):
        # p2m_graph_destroy calls hipFree, which is illegal while ANY stream of the thread captures (it invalidates the
        # capture): the cyclic garbage collector may run this finalizer at any allocation - also inside a `with
        # torch.cuda.graph(...)` block (round 5: GraphedInference re-captures failed with hipErrorStreamCaptureInvalidated in
        # long test processes full of collectable models).  During a capture the handle is parked and freed later.
        try:
            h = getattr(self, "handle", None)
            if h:
                self.handle = None
                if _capturing():
                    _deferred_destroy.append(h)
                else:
                    _lib.hip().p2m_graph_destroy(h)
        except Exception:
            pass


_deferred_destroy = []      # graph handles whose finalizer ran during a stream capture (DeviceGraph.__del__)


def drain_deferred_destroys():
    """Free the graph handles parked by finalizers that ran inside a stream capture (no-op while capturing)."""
    if _deferred_destroy and not _capturing():
        while _deferred_destroy:
            try:
                _lib.hip().p2m_graph_destroy(_deferred_destroy.pop())
            except Exception:
                pass


class capture_guard:
    """Context manager around a stream capture (infer.GraphedInference, train.GraphedTrainStep): collects garbage BEFORE the
    capture and keeps the cyclic collector off inside it, so that no finalizer - ours (hipFree) or torch's (graphs, events,
    streams of dead objects) - runs API calls that are illegal during a capture; parked handles are freed afterwards."""

    def __enter__(self):
        import gc
        gc.collect()
        drain_deferred_destroys()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *a):
        import gc
        if self._was:
            gc.enable()
        drain_deferred_destroys()
        return False


def class_representatives(V, fake_ids, depth):
    """rep_of[V] for a level with `depth` un-pool steps above it in the stack: the maximal aligned runs of 2^j <= 2^depth
    all-fake rows are the descendants of ONE fake vertex j levels up, hence identical (include/p2m.h "classes")."""
    fake = np.zeros(V, dtype=bool)
    fake[np.asarray(fake_ids, dtype=np.int64)] = True
    rep = np.arange(V, dtype=np.int32)
    for j in range(1, depth + 1):
        bs = 1 << j
        if V % bs:
            break
        for blk in np.where(fake.reshape(V // bs, bs).all(axis=1))[0]:
            rep[blk * bs:(blk + 1) * bs] = blk * bs
    return rep, fake


class GraphCache:
    """Per-device DeviceGraph lists for a list of scipy Laplacians (thread-safe: nn.DataParallel
    calls forward from one thread per GPU on replicas that share this object).
    class_plan: optional callable(list of DeviceGraph) -> {graph index: depth} deciding, once the handles exist, on which
    levels runs of identical fake rows are declared (the caller knows the un-pool chain)."""

    def __init__(self, laplacians, class_plan=None):
        self.laplacians = list(laplacians)
        self.class_plan = class_plan
        self._per_device = {}
        self._lock = threading.Lock()

    def on(self, device):
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        g = self._per_device.get(key)
        if g is None:
            with self._lock:
                g = self._per_device.get(key)
                if g is None:
                    g = [DeviceGraph(L, torch.device("cuda", key)) for L in self.laplacians]
                    if self.class_plan is not None and CLASSES:
                        for gi, depth in (self.class_plan(g) or {}).items():
                            rep, _ = class_representatives(g[gi].V, g[gi].fake_ids_host(), depth)
                            g[gi].set_classes(rep)
                    self._per_device[key] = g
        return g


# ---------------------------------------------------------------------------------------------
# thin wrappers (one per C-ABI entry point)
# ---------------------------------------------------------------------------------------------
def stats_tile_rows():
    return int(_lib.hip().p2m_stats_tile_rows())


def cheb_basis_fwd(g, X, B, F, in_shift):
    M = B * g.V
    T1 = torch.empty((M, F), device=X.device, dtype=torch.float32)
    T2 = torch.empty((M, F), device=X.device, dtype=torch.float32)
    # bytes this launch moves (the un-pooled input is read at the coarse resolution), SURVEY 8(d): 4*K*V*Fin per sample
    with _timed("cheb_basis_fwd", (4.0 * M * F * (2.0 + 1.0 / (1 << in_shift)), 12.0 * M * F)):
        check(_lib.hip().p2m_cheb_basis_fwd(g.handle, _p(_req(X, "X")), _p(T1), _p(T2), B, F, in_shift, _stream()),
              "p2m_cheb_basis_fwd")
    return T1, T2


def cheb_basis_bwd(g, d0, d1, d2, resid, B, F, out_shift):
    dX = torch.empty((B * (g.V >> out_shift), F), device=d0.device, dtype=torch.float32)
    nbytes = 4.0 * B * g.V * F * ((4.0 if resid is not None else 3.0) + 1.0 / (1 << out_shift))
    with _timed("cheb_basis_bwd", nbytes):
        check(_lib.hip().p2m_cheb_basis_bwd(g.handle, _p(_req(d0, "d0")), _p(_req(d1, "d1")), _p(_req(d2, "d2")),
                                            _p(resid if resid is None else _req(resid, "resid")), _p(dX), B, F,
                                            out_shift, _stream()), "p2m_cheb_basis_bwd")
    return dX


def cheb_combine_small(g, P, nc, bias, B):
    """Y = P0 + L P1 + L2 P2 (+bias) on nc <= 4 columns; P: [B*V, ldp]."""
    M = B * g.V
    Y = torch.empty((M, nc), device=P.device, dtype=torch.float32)
    with _timed("cheb_small", 4.0 * M * (P.shape[1] + nc)):
        check(_lib.hip().p2m_cheb_combine_small(g.handle, _p(_req(P, "P")), P.shape[1], nc,
                                                _p(bias if bias is None else _req(bias, "bias")), _p(Y), B, _stream()),
              "p2m_cheb_combine_small")
    return Y


def cheb_combine_small_real(g, P, nc, bias, B, out_index=None, out_rows=0, scale=1.0):
    """Inference: the combine over the real vertices only.  out_index ([V] int32 on the device, -1 = drop) stores vertex
    v at row out_index[v] of a [B, out_rows, nc] tensor (mesh-model order) times `scale`; without it the result is
    [B*V, nc] with zeros at the padding vertices."""
    if out_index is None:
        Y = torch.zeros((B * g.V, nc), device=P.device, dtype=torch.float32)
    else:
        Y = torch.empty((B, out_rows, nc), device=P.device, dtype=torch.float32)
    with _timed("cheb_small", 4.0 * B * g.n_real * (P.shape[1] + nc)):
        check(_lib.hip().p2m_cheb_combine_small_real(g.handle, _p(_req(P, "P")), P.shape[1], nc,
                                                     _p(bias if bias is None else _req(bias, "bias")), _p(Y), B,
                                                     _p(out_index), int(out_rows), float(scale), _stream()),
              "p2m_cheb_combine_small_real")
    return Y


def cheb_expand_small(g, G, nc, lde, B):
    """E = [G | L G | L2 G | 0] with rows lde wide; G: [B*V, nc]."""
    M = B * g.V
    E = torch.empty((M, lde), device=G.device, dtype=torch.float32)
    with _timed("cheb_small", 4.0 * M * (lde + nc)):
        check(_lib.hip().p2m_cheb_expand_small(g.handle, _p(_req(G, "G")), nc, _p(E), lde, B, _stream()),
              "p2m_cheb_expand_small")
    return E


# ---- fake-vertex split (row-set launches) ---------------------------------------------------
def cheb_basis_fwd_real(g, X, B, F, in_shift, in_act=None):
    """Basis planes of the REAL vertices only, compact [B*n_real, F].  in_act = (scale[F], shift[F]): activation on load - X
    is a raw conv output, the planes are those of relu(X * scale + shift) (tile-plan kernel only)."""
    T1 = torch.empty((B * g.n_real, F), device=X.device, dtype=torch.float32)
    T2 = torch.empty((B * g.n_real, F), device=X.device, dtype=torch.float32)
    # bytes moved: real rows only (the fake rows' planes are folded into the effective weight); SURVEY 8(d) counts all V
    with _timed("cheb_basis_fwd", (4.0 * B * g.n_real * F * (2.0 + 1.0 / (1 << in_shift)), 12.0 * B * g.V * F)):
        check(_lib.hip().p2m_cheb_basis_fwd_real(g.handle, _p(_req(X, "X")), _p(T1), _p(T2), B, F, in_shift,
                                                 _p(None if in_act is None else _req(in_act[0], "act_scale")),
                                                 _p(None if in_act is None else _req(in_act[1], "act_shift")), _stream()),
              "p2m_cheb_basis_fwd_real")
    return T1, T2


# Arithmetic of the dense contractions (include/p2m.h P2M_ARITH_*), all of them the fp32 contraction with fp32
# accumulation: "f16x2" = operands scaled by a power of two from their amax word and cut into 2 fp16 slices, 3 slice
# products on the fp16 matrix pipe (22-bit operands); "bf16x3" = operands cut exactly into 3 bf16 slices, 6 slice products
# (error vs float64 <= the native kernel's, tests/test_gpu_ops.py::test_bf16x3_error_is_fp32_class); "f32" = native f32
# MFMA (the independent kernel set of the parity tests).
# Default f16x2: same parity maxima against the float64 oracle as bf16x3 (tests/test_gpu_parity_full.py) at half the
# matrix-core work and two thirds of the LDS traffic - train step 40.1 vs 44.1 ms on the same MI355X.
GEMM_ARITH = _os.environ.get("P2M_GEMM_ARITH", "f16x2")
if GEMM_ARITH not in ("f32", "bf16x3", "f16x2"):
    raise ValueError(f"P2M_GEMM_ARITH must be f32, bf16x3 or f16x2, not {GEMM_ARITH!r}")


# ---- amax words (include/p2m.h, P2M_ARITH_F16X2) ----------------------------------------------------------------
# A uint32 in device memory bounding max |x| of one tensor.  The producer of a tensor attaches the word to the tensor
# OBJECT (`t._p2m_amax`); a consumer that finds none computes it with one extra pass (amax_of).  Words come zeroed from
# a per-device chunk that is renewed at the start of every forward (inside a stream capture the zeroing is then part of
# the graph, and a forward never zeroes words a pending backward still reads).
_amax_chunks = {}


def f16x2():
    return GEMM_ARITH == "f16x2"


def amax_begin_step(device):
    """Renew the device's chunk of amax words: called at the start of every network forward that hands words out (a step
    then draws from one chunk; the stand-alone ChebConv op just draws from whatever chunk is current)."""
    _amax_chunks.pop(torch.device(device).index, None)
    if _deferred_destroy:
        drain_deferred_destroys()


def _capturing():
    """0 when the current stream is not capturing, else the IDENTITY of the capture it records into (the runtime's capture
    id, p2m_stream_capture_id): two captures taken back to back are two different states."""
    try:
        if not torch.cuda.is_current_stream_capturing():
            return 0
    except Exception:
        return 0
    cid = ctypes.c_uint64(0)
    check(_lib.hip().p2m_stream_capture_id(_stream(), ctypes.byref(cid)), "p2m_stream_capture_id")
    return int(cid.value) or -1


def new_amax(device):
    """One zeroed amax word.  A chunk belongs to the capture it was created in (0: none): a chunk made before a stream
    capture is never re-zeroed by the graph's replays (its words would only grow), and one made INSIDE a capture is that
    graph's private memory (zeroed by every replay of THAT graph, uninitialised before the first, gone when the graph is
    destroyed) - so the chunk is dropped whenever the capture identity changes, also from one capture straight to the next
    (round 6: it used to be keyed on a boolean).
    A chunk is zeroed on the stream that created it; any other stream that draws a word from it first waits for that
    (inside a capture the record / wait pair becomes a dependency edge of the graph)."""
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    with torch.cuda.device(key):
        cap = _capturing()
    ent = _amax_chunks.get(key)
    if ent is None or ent[1] >= ent[0].numel() or ent[2] != cap:
        with torch.cuda.device(key):
            buf = torch.zeros(256, dtype=torch.int32, device=torch.device("cuda", key))
            st = torch.cuda.current_stream()
            # also inside a stream capture: event record / wait are capturable and become an EDGE of the graph - a chunk
            # that rolls over on the weight-gradient side stream is then zeroed before the main stream's later atomics on it
            ev = torch.cuda.Event()
            ev.record(st)
        ent = [buf, 0, cap, ev, {st.cuda_stream}]
        _amax_chunks[key] = ent
    elif ent[3] is not None:
        st = torch.cuda.current_stream(torch.device("cuda", key))
        if st.cuda_stream not in ent[4]:
            st.wait_event(ent[3])
            ent[4].add(st.cuda_stream)
    w = ent[0][ent[1]:ent[1] + 1]
    ent[1] += 1
    return w


def tag_amax(t, word):
    """Attach the amax word of t's CURRENT contents (the producing kernel wrote both).  The tag remembers the tensor's
    storage and version: a caller that later refills the tensor through torch (in place: the version moves; or rebinds
    .data: the pointer moves) gets a fresh bound from amax_of instead of the stale word."""
    t._p2m_amax = word
    t._p2m_amax_of = (t.data_ptr(), t._version)
    return t


def _tag_valid(t):
    return getattr(t, "_p2m_amax", None) is not None and getattr(t, "_p2m_amax_of", None) == (t.data_ptr(), t._version)


def view_tagged(t, *shape):
    """t.view(*shape) that keeps t's amax word (a view is a new tensor object)."""
    v = t.view(*shape)
    if _tag_valid(t):
        tag_amax(v, t._p2m_amax)
    return v


def amax_of(t, g=None, B=None, row_set=0):
    """The amax word of t (None unless GEMM_ARITH is f16x2): the one its producer attached - if the tensor has not been
    modified through torch since -, else computed now - over the rows of `row_set` of level g when given (0: every row that
    holds data), else over the whole tensor."""
    if not f16x2() or t is None or t.shape[-1] % 4 != 0:    # (widths that are not MFMA shapes take the scalar kernels)
        return None
    if _tag_valid(t):
        return t._p2m_amax
    w = new_amax(t.device)
    if g is not None:
        F = t.shape[-1]
        if t.numel() != B * g.V * F:
            raise P2MError("amax_of: the tensor is not a [B, V, F] tensor of this level (un-pooled operands must come "
                           "tagged from their producer)")
        check(_lib.hip().p2m_amax_rows(g.handle, row_set, _p(_req(t, "x")), B, F, _p(w), _stream()), "p2m_amax_rows")
    else:
        check(_lib.hip().p2m_amax(_p(_req(t, "x")), t.numel(), _p(w), _stream()), "p2m_amax")
    tag_amax(t, w)
    return w


# ---- per-step cache of derived weight operands ---------------------------------------------------------------
# Packed / transposed / pre-split / fake-vertex-effective copies of a weight are constant between optimizer steps, but the
# network needs them in the forward AND in the backward of every conv (~85 tiny launches per step).  They are cached
# per (layer, kind, device) and rebuilt when the parameter changes: torch updates bump Tensor._version; the flat
# optimizers (optim.py) update the buffer through a raw pointer and bump WEIGHT_EPOCH instead.
WEIGHT_EPOCH = 0


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


class WeightCache:
    def __init__(self):
        self._d = {}
        self.builds = 0          # builder calls so far

    def get(self, key, W, builder):
        """W: the source tensor, or a tuple of source tensors."""
        Ws = W if isinstance(W, tuple) else (W,)
        tag = tuple((t.data_ptr(), t._version) for t in Ws) + (WEIGHT_EPOCH,)
        key = (key, Ws[0].device.index)
        hit = self._d.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        val = builder()
        self.builds += 1
        self._d[key] = (tag, val)
        return val

    def clear(self):
        self._d.clear()


def _amax_planes(planes, g=None, B=None):
    """One word bounding SEVERAL operand planes that are not known to derive from one tagged tensor (op-level callers;
    the network passes the source tensor's word plus headroom bits instead): planes[0] over the data rows of level g when
    given, the others (compact planes) whole."""
    if len(planes) == 1:
        return amax_of(planes[0], g, B)
    w = new_amax(planes[0].device)
    for i, t in enumerate(planes):
        if i == 0 and g is not None:
            check(_lib.hip().p2m_amax_rows(g.handle, 0, _p(_req(t, "x")), B, t.shape[-1], _p(w), _stream()), "p2m_amax_rows")
        else:
            check(_lib.hip().p2m_amax(_p(_req(t, "x")), t.numel(), _p(w), _stream()), "p2m_amax")
    return w


def gemm_kernel_name():
    """Name prefix of the plane-contraction kernel the current arithmetic selects (rocprof kernel names start with it)."""
    return "k_gemm_planes<" if GEMM_ARITH == "f32" else "k_gemm_planes_ws"


def arith_code():
    """P2M_ARITH_* value of include/p2m.h for the current GEMM_ARITH."""
    return {"f32": 0, "bf16x3": 1, "f16x2": 2}[GEMM_ARITH]


def weight_split(Bm, amax=None, bits=0):
    """Pre-split, k-contiguous copy of a [K, N] weight operand for the slice contractions (None in f32 mode).
    f16x2: amax = the amax word of the tensor Bm derives from, which bounds Bm after `bits` binades (param_amax,
    eff_bits); default: Bm's own maximum, computed by one more small launch."""
    if GEMM_ARITH == "f32":
        return None
    K, N = Bm.shape
    lib = _lib.hip()
    Bx = torch.empty((int(lib.p2m_weight_split_elems(K, N, arith_code())),), device=Bm.device, dtype=torch.int16)
    check(lib.p2m_weight_split(_p(_req(Bm, "B")), K, N, arith_code(), _p(amax), int(bits), _p(Bx), _stream()),
          "p2m_weight_split")
    return Bx


def param_amax(W):
    """f16x2: one amax word per parameter tensor - its packed / transposed / effective copies are bounded by it."""
    if not f16x2():
        return None
    w = new_amax(W.device)
    check(_lib.hip().p2m_amax(_p(_req(W, "weight")), W.numel(), _p(w), _stream()), "p2m_amax")
    return w


def eff_bits(a, b):
    """binades by which W0 + a W1 + b W2 (weight_eff) can exceed max |W|"""
    return int(np.ceil(np.log2(1.0 + abs(a) + abs(b)) - 1e-12))


def gemm_planes_rows(g, row_set, B, A, Ka, a0_shift, compact, Bm, bias, addend, C, N, stats=False, Bx=None, act=None,
                     amax=None, amax_bits=0, amax_out=None, in_act=None):
    """Row-set contraction into the rows of C selected by row_set (1 real, 2 fake, 3 / 4 the paired sets).  Returns stats
    or None.  Bx: the pre-split copy of Bm (weight_split) when the caller has it cached.  f16x2: amax = the word
    bounding the A planes after amax_bits binades (default: A[0]'s own); amax_out: a zeroed word that receives the
    bound of what is stored.  in_act = (scale[Ka], shift[Ka]): activation on load of plane 0 (A[0] is a raw conv output);
    amax must then bound the activated operand."""
    n = g.set_size(row_set)
    if in_act is not None and amax is None and f16x2():
        raise P2MError("gemm_planes_rows: activation on load needs the amax word of the activated operand (act_bound)")
    if f16x2() and amax is None:
        amax = _amax_planes(A, g if (a0_shift == 0 and row_set <= 2) else None, B)
    st = None
    if stats:
        tps = int(_lib.hip().p2m_rows_tiles_per_sample(g.handle, row_set))
        st = torch.empty((B * tps, 2, N), device=C.device, dtype=torch.float32)
    a = [_p(_req(t, "A plane")) for t in A] + [None] * (3 - len(A))
    fl = 2.0 * B * n * len(A) * Ka * N
    # algorithmic HBM bytes: every A plane row once, the output once (weights come from L2)
    with _timed("gemm_planes_mfma", (fl, fl, 4.0 * B * n * (len(A) * Ka + N))):
        check(_lib.hip().p2m_gemm_planes_rows(g.handle, row_set, B, a[0], a[1], a[2], len(A), Ka, a0_shift,
                                              int(compact), _p(_req(Bm, "B")),
                                              _p(Bx if Bx is not None else weight_split(Bm)), arith_code(), _p(amax),
                                              int(amax_bits),
                                              _p(bias if bias is None else _req(bias, "bias")),
                                              _p(addend if addend is None else _req(addend, "addend")), _p(C), N,
                                              _p(st), _p(None if act is None else act[0]),
                                              _p(None if act is None else act[1]), int(bool(act and act[2])),
                                              _p(amax_out), _p(None if in_act is None else _req(in_act[0], "in_scale")),
                                              _p(None if in_act is None else _req(in_act[1], "in_shift")), _stream()),
              "p2m_gemm_planes_rows")
    return st


_side_streams = {}
_side_lock = threading.Lock()


def side_stream(device, which=0):
    """Helper streams per device (0 = weight-gradient GEMMs of the backward).  Locked: nn.DataParallel runs one
    thread per GPU through here."""
    key = (torch.device(device).index, which)
    st = _side_streams.get(key)
    if st is None:
        with _side_lock:
            st = _side_streams.get(key)
            if st is None:
                st = torch.cuda.Stream(device=device)
                _side_streams[key] = st
    return st


class _ConvWeightsDesc(ctypes.Structure):       # == p2m_conv_weights (include/p2m.h)
    _fields_ = [("W", _vp), ("Fout", ctypes.c_int32), ("Fin", ctypes.c_int32), ("fake_a", ctypes.c_float),
                ("fake_b", ctypes.c_float), ("eff_bits", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("Bx_f", _vp), ("Bx_ef", _vp), ("Bx_b", _vp), ("Bx_eb", _vp), ("amax", _vp)]


class ConvWeightSet:
    """The four slice images (forward / backward, real / padding rows) of EVERY split conv of a network, refreshed by
    p2m_conv_weights_prepare in two launches per optimizer step instead of eight per layer (include/p2m.h).  Buffers and the
    device-side descriptor array are allocated once per device; `refresh` re-uploads the descriptors only when a weight
    moved.  entries: list of (key, weight tensor, fake_a, fake_b)."""

    def __init__(self, entries, device):
        self.device = torch.device(device)
        self.keys = [k for k, _, _, _ in entries]
        self.weights = [w for _, w, _, _ in entries]
        self.coef = [(float(a), float(b)) for _, _, a, b in entries]
        lib = _lib.hip()
        code = arith_code()
        self.arith = code
        self.images = {}
        for k, w in zip(self.keys, self.weights):
            Fout, Fin = w.shape[0], w.shape[1] // 3
            shapes = ((3 * Fin, Fout), (Fin, Fout), (3 * Fout, Fin), (Fout, Fin))
            self.images[k] = tuple(torch.zeros((int(lib.p2m_weight_split_elems(K, N, code)),), device=self.device,
                                               dtype=torch.int16) for K, N in shapes)
        self.words = torch.zeros((len(entries),), device=self.device, dtype=torch.int32)
        self.desc = None
        self._ptrs = None
        self.tag = None

    def _upload(self):
        arr = (_ConvWeightsDesc * len(self.keys))()
        for i, (k, w, (a, b)) in enumerate(zip(self.keys, self.weights, self.coef)):
            f, ef, bw, eb = self.images[k]
            arr[i] = _ConvWeightsDesc(_req(w, "weight").data_ptr(), w.shape[0], w.shape[1] // 3, a, b, eff_bits(a, b), 0,
                                      f.data_ptr(), ef.data_ptr(), bw.data_ptr(), eb.data_ptr(),
                                      self.words[i:i + 1].data_ptr())
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.desc = host.to(self.device)
        self._ptrs = [w.data_ptr() for w in self.weights]

    def refresh(self):
        """Recompute every image from the current weights (no-op when they have not changed since the last call)."""
        tag = tuple((w.data_ptr(), w._version) for w in self.weights) + (WEIGHT_EPOCH,)
        if tag == self.tag:
            return
        if self.desc is None or self._ptrs != [w.data_ptr() for w in self.weights]:
            self._upload()
        check(_lib.hip().p2m_conv_weights_prepare(_p(self.desc), len(self.keys), self.arith, _stream()),
              "p2m_conv_weights_prepare")
        self.tag = tag

    def fwd(self, key, dummy):
        """(Bx, We, Wex) as split_operands returns them; `dummy`: any fp32 tensor standing in for the fp32 operands the
        slice kernels never read."""
        f, ef, _, _ = self.images[key]
        return f, dummy, ef

    def bwd(self, key, dummy):
        _, _, b, eb = self.images[key]
        return b, dummy, eb


def split_operands(Bm, Ka, N, fake_a, fake_b, amax=None):
    """Derived operands of one split contraction: (Bx, We, Wex) = pre-split Bm, the fake-vertex effective weight
    W0 + a*W1 + b*W2 and its pre-split copy.  amax: the amax word of the parameter Bm is a permutation of (f16x2)."""
    We = weight_eff(Bm, Ka, N, fake_a, fake_b)
    return weight_split(Bm, amax), We, weight_split(We, amax, eff_bits(fake_a, fake_b) if amax is not None else 0)


def cheb_basis_pair(g, G, B, F):
    """S L g and S L2 g (S = pair-sum) over the coarse vertices with a real child, compact [B*n_pair_real, F]."""
    P1 = torch.empty((B * g.n_pair_real, F), device=G.device, dtype=torch.float32)
    P2 = torch.empty((B * g.n_pair_real, F), device=G.device, dtype=torch.float32)
    with _timed("cheb_basis_fwd", (4.0 * F * B * (g.V + 2.0 * g.n_pair_real), 12.0 * B * g.V * F)):
        check(_lib.hip().p2m_cheb_basis_pair(g.handle, _p(_req(G, "G")), _p(P1), _p(P2), B, F, _stream()),
              "p2m_cheb_basis_pair")
    return P1, P2


def tile_gemm_ok(g, plan, Ka, N, want_planes=False, B=None):
    """True when the real rows of this conv take the basis-inside-the-contraction kernel (see TILE_GEMM)."""
    if TILE_GEMM in ("0", False) or GEMM_ARITH == "f32":
        return False
    if not _lib.hip().p2m_cheb_tile_gemm_supported(g.handle, plan, Ka, N):
        return False
    if TILE_GEMM in ("1", True):
        return True
    if plan == 2:
        return False
    if _lib.hip().p2m_cheb_tile_gemm_mg(arith_code(), N):
        # the gather runs on the matrix cores (k_cheb_mg_gemm: f16x2; bf16x3 with P2M_MG_EXACT=1): also with the planes
        # written out, at any batch (f16x2: B = 64 inference 3.51 vs 3.96 ms, B = 8 1.63 vs 1.76)
        return g.n_real >= TILE_GEMM_MG_MIN_ROWS
    if GEMM_ARITH == "bf16x3":
        # Round 5 (profiles/r05_d_probe_mg0.txt, B = 256): in the exact arithmetic the VALU-gather tile kernel beats basis
        # kernel + plane contraction on EVERY own / un-pooled plan of the split levels, N = 256 and planes out included
        # (finest 128 -> 128: 1.85 vs 2.51 ms forward, 2.08 vs 2.51 with the planes; 1 472-vertex level 256 -> 256: 0.89 vs 1.08):
        # six slice products per MFMA step hide the gather's VALU work that the three of f16x2 do not.
        return g.n_real >= TILE_GEMM_MG_MIN_ROWS and (B is None or B >= TILE_GEMM_MIN_BATCH)
    return not want_planes and g.n_real >= TILE_GEMM_MIN_ROWS and (B is None or B >= TILE_GEMM_MIN_BATCH)


def tile_bnr_slots(g, plan, N, B):
    """Partial-sum slots of the BatchNorm-backward reduction fused into p2m_cheb_tile_gemm (bnr=...), 0: not available for
    the current arithmetic / this width (include/p2m.h)."""
    if not BN_FUSE or GEMM_ARITH == "f32":
        return 0
    return int(_lib.hip().p2m_cheb_tile_gemm_bnr_slots(g.handle, plan, arith_code(), N, B))


def cheb_tile_gemm(g, plan, X, A0, Ka, Bx, bias, addend, C, N, B, stats=False, want_planes=False, act=None,
                   amax=None, amax_out=None, in_act=None, bnr=None):
    """C[rows of the plan] = [A0 | L X | L2 X] W (+bias)(+addend) in one kernel (include/p2m.h).  Returns
    (stats [B*ntiles, 2, N] or None, (E1, E2) compact planes or None).  f16x2: amax = the word bounding X and A0 (default:
    X's own); amax_out: a zeroed word that receives the bound of what is stored.  in_act = (scale[Ka], shift[Ka]):
    activation on load - X / A0 hold a raw conv output y, the operand is relu(y * scale + shift); amax must bound THAT.
    bnr = (y [B*c_rows, N], co [4, N], part [slots, 2, N]): the BatchNorm-backward reduction of the layer in front over the
    rows this launch stores, written to `part` (tile_bnr_slots(g, plan, N, B) slots)."""
    nset = g.n_pair_real if plan == 2 else g.n_real
    if in_act is not None and amax is None and f16x2():
        raise P2MError("cheb_tile_gemm: activation on load needs the amax word of the activated operand (act_bound)")
    if f16x2() and amax is None:
        amax = amax_of(X, g if plan != 1 else None, B)
    st = torch.empty((B * g.plan_tiles[plan], 2, N), device=C.device, dtype=torch.float32) if stats else None
    E1 = torch.empty((B * nset, Ka), device=C.device, dtype=torch.float32) if want_planes else None
    E2 = torch.empty((B * nset, Ka), device=C.device, dtype=torch.float32) if want_planes else None
    fl = 2.0 * B * nset * 3 * Ka * N
    # algorithmic HBM bytes (fused rule): the input rows once, the output once (+ the planes when asked for)
    nbytes = 4.0 * B * nset * (Ka + N + (2 * Ka if want_planes else 0) + (N if addend is not None else 0))
    with _timed("cheb_tile_gemm", (fl, fl, nbytes)):
        check(_lib.hip().p2m_cheb_tile_gemm(g.handle, plan, _p(_req(X, "X")), _p(_req(A0, "A0")), Ka, _p(Bx),
                                            arith_code(), _p(amax),
                                            _p(bias if bias is None else _req(bias, "bias")),
                                            _p(addend if addend is None else _req(addend, "addend")), _p(C), N, _p(st),
                                            _p(E1), _p(E2), _p(None if act is None else act[0]),
                                            _p(None if act is None else act[1]), int(bool(act and act[2])),
                                            _p(amax_out), _p(None if in_act is None else _req(in_act[0], "in_scale")),
                                            _p(None if in_act is None else _req(in_act[1], "in_shift")),
                                            _p(None if bnr is None else _req(bnr[0], "bnr_y")),
                                            _p(None if bnr is None else _req(bnr[1], "bnr_co")),
                                            _p(None if bnr is None else bnr[2]), B, _stream()),
              "p2m_cheb_tile_gemm")
    return st, ((E1, E2) if want_planes else None)


def bn_finalize_tiles(g, plan, B, st_real, st_fake, gamma, beta, running_mean, running_var, momentum, eps):
    N = gamma.shape[0]
    co = torch.empty((4, N), device=gamma.device, dtype=torch.float32)
    check(_lib.hip().p2m_bn_finalize_tiles(g.handle, plan, _p(st_real), _p(st_fake), B, _p(_req(gamma, "bn.weight")),
                                           _p(_req(beta, "bn.bias")), _p(running_mean), _p(running_var),
                                           float(momentum), float(eps), _p(co[0]), _p(co[1]), _p(co[2]), _p(co[3]), N,
                                           _stream()), "p2m_bn_finalize_tiles")
    return co


def conv_pair(g, B, Gy, Ka, Bm, addend, C, N, operands, P0=None, amax_out=None):
    """Backward contraction of an un-pooled conv at the coarse resolution (include/p2m.h "paired operator"):
    C[B*V/2, N] = [S g | S L g | S L2 g] Bm (+ addend).  Returns the planes (P0 full, P1c, P2c).
    P0: S g when the caller already has it (by-product of the BatchNorm backward)."""
    if P0 is None:
        P0 = pair_sum(Gy, B * (g.V // 2), Ka, classes=g)
    Bx, We, Wex = operands
    ga = amax_of(Gy, g, B)            # |S g| <= 2 max |g|, |S L g|, |S L2 g| <= 2^(plane_bits + 1) max |g|
    if tile_gemm_ok(g, 2, Ka, N, want_planes=True, B=B):
        # planes S L g, S L2 g formed inside the contraction; written out (compact) only for the weight gradient
        _, (P1c, P2c) = cheb_tile_gemm(g, 2, Gy, P0, Ka, Bx, None, addend, C, N, B, want_planes=True, amax=ga,
                                       amax_out=amax_out)
    else:
        P1c, P2c = cheb_basis_pair(g, Gy, B, Ka)
        gemm_planes_rows(g, 3, B, [P0, P1c, P2c], Ka, 0, True, Bm, None, addend, C, N, False, Bx=Bx, amax=ga,
                         amax_bits=g.plane_bits + 1, amax_out=amax_out)
    gemm_planes_rows(g, 4, B, [P0], Ka, 0, False, We, None, addend, C, N, False, Bx=Wex, amax=ga, amax_bits=1,
                     amax_out=amax_out)
    return P0, P1c, P2c


BASIS_TILED = _os.environ.get("P2M_BASIS_TILED", "1") == "1"     # (the library reads the same variable: INTEGRATION.md section 7)


def fold_act_ok(g, Ka, N, B, narrow=False):
    """True when a conv with input width Ka and output width N on the (split) level g can take its input as a RAW conv
    output with the activation applied on load (FOLD_ACT; both slice arithmetics since round 5 - bf16x3 needs no amax
    bound).  Three consumer forms implement it: the tile kernels on the level's own plan (N <= 128: the matrix-core gather
    in f16x2, the VALU gather in bf16x3); the LDS-staged basis kernel + plane contraction (needs the level's tile plan, Ka <=
    256); the narrow final conv's projection (row-set contractions: needs declared classes).  The fake-vertex rows and every
    weight gradient read the raw tensor through the same on-load activation."""
    if not (FOLD_ACT and GEMM_ARITH != "f32" and g.split and Ka <= 256 and Ka % 32 == 0):
        return False
    if narrow:
        return bool(g.classes)
    if tile_gemm_ok(g, 0, Ka, N, False, B=B):
        return True
    return bool(BASIS_TILED and g.plan_tiles[0] > 0 and (Ka in (32, 64) or Ka % 128 == 0) and N % 32 == 0
                and not tile_gemm_ok(g, 0, Ka, N, False, B=B))


def act_bound(scale, shift, y_amax, word):
    """word = max(word, bound of relu(y * scale + shift) given the amax word of y) (include/p2m.h p2m_act_bound)."""
    check(_lib.hip().p2m_act_bound(_p(_req(scale, "scale")), _p(_req(shift, "shift")), scale.shape[0], _p(y_amax),
                                   _p(word), _stream()), "p2m_act_bound")
    return word


def conv_split(g, B, X, Ka, a0_shift, Bm, bias, addend, C, N, fake_a, fake_b, stats=False, operands=None, want_planes=True,
               amax_out=None, in_act=None, bnr=None):
    """One split contraction: the real-vertex rows [X | L X | L2 X] Bm (K = 3*Ka), then the fake-vertex GEMM (K = Ka,
    W0 + a*W1 + b*W2).  All on the current stream: running the fake-vertex GEMM or half of the batch's basis on a side
    stream was measured neutral (DESIGN.md "Streams").  Returns (T1c, T2c, st_real, st_fake, tiled): the compact basis
    planes of the real vertices (None unless want_planes), the BatchNorm partials of the two launches, and whether
    st_real is in per-(sample, tile) form (p2m_bn_finalize_tiles, plan = a0_shift) or per 128-row tile
    (p2m_bn_finalize_split).  in_act = (scale, shift) - activation on load (fold_act_ok): X is the RAW output of the previous
    conv, tagged with the amax word that bounds relu(X * scale + shift), and every reader applies that on load."""
    Bx, We, Wex = operands if operands is not None else split_operands(Bm, Ka, N, fake_a, fake_b)
    xa = amax_of(X, g if a0_shift == 0 else None, B)
    if in_act is not None:
        # X IS the raw output of the previous conv (tagged with the bound of its activated form); every reader applies
        # relu(y * scale + shift) on load: tile kernel / basis kernel + plane 0 of the contraction / fake-row contraction
        if a0_shift != 0 or want_planes:
            raise P2MError("conv_split: activation on load needs the level's own plan and no planes out (fold_act_ok)")
        if tile_gemm_ok(g, 0, Ka, N, False, B=B):
            st1, _ = cheb_tile_gemm(g, 0, X, X, Ka, Bx, bias, addend, C, N, B, stats=stats, amax=xa, amax_out=amax_out,
                                    in_act=in_act)
            tiled = True
        else:
            T1c, T2c = cheb_basis_fwd_real(g, X, B, Ka, 0, in_act=in_act)
            st1 = gemm_planes_rows(g, 1, B, [X, T1c, T2c], Ka, 0, True, Bm, bias, addend, C, N, stats, Bx=Bx, amax=xa,
                                   amax_bits=g.plane_bits, amax_out=amax_out, in_act=in_act)
            tiled = False
        st2 = gemm_planes_rows(g, 2, B, [X], Ka, 0, False, We, bias, addend, C, N, stats, Bx=Wex, amax=xa,
                               amax_out=amax_out, in_act=in_act)
        return None, None, st1, st2, tiled
    if bnr is not None and not (in_act is None and tile_gemm_ok(g, a0_shift, Ka, N, want_planes, B=B)):
        raise P2MError("conv_split: bnr needs the tile kernel (tile_bnr_slots > 0 and tile_gemm_ok)")
    if tile_gemm_ok(g, a0_shift, Ka, N, want_planes, B=B):
        st1, planes = cheb_tile_gemm(g, a0_shift, X, X, Ka, Bx, bias, addend, C, N, B, stats=stats,
                                     want_planes=want_planes, amax=xa, amax_out=amax_out, bnr=bnr)
        st2 = gemm_planes_rows(g, 2, B, [X], Ka, a0_shift, False, We, bias, addend, C, N, stats, Bx=Wex, amax=xa,
                               amax_out=amax_out)
        T1c, T2c = planes if planes is not None else (None, None)
        return T1c, T2c, st1, st2, True
    T1c, T2c = cheb_basis_fwd_real(g, X, B, Ka, a0_shift)
    st1 = gemm_planes_rows(g, 1, B, [X, T1c, T2c], Ka, a0_shift, True, Bm, bias, addend, C, N, stats, Bx=Bx, amax=xa,
                           amax_bits=g.plane_bits, amax_out=amax_out)
    st2 = gemm_planes_rows(g, 2, B, [X], Ka, a0_shift, False, We, bias, addend, C, N, stats, Bx=Wex, amax=xa,
                           amax_out=amax_out)
    return T1c, T2c, st1, st2, False


# Blocks a weight-gradient launch should have at least (512 block slots: 2 per CU), and the row-set policy that follows from it:
# a launch whose one-chunk-per-sample form has >= 2 TN_TARGET_BLOCKS blocks takes floor(blocks / TN_TARGET_BLOCKS) WHOLE samples
# per chunk (include/p2m.h, p2m_gemm_tn_rows with splits < 0).  Every chunk costs a partial buffer P[chunk][Ka][N] that the
# contraction writes and the unpack reads: at one chunk per sample the 256 x 768 gradients of the levels with 544 - 1 021 rows per
# sample moved 0.4 - 0.7 of their operand bytes again as partials.  Round 6, same box, bf16x3 step: one chunk per sample at
# 768 blocks 43.49 ms; whole samples per chunk at 768: 43.05; at 512: 42.28 (42.10); at 384: 42.00 (41.70); at 320 / 256:
# 42.0 / 42.08 (`profiles/r06_tn_samples_per_chunk_ab.txt`).  The kernel ALONE is slower with fewer blocks (weight gradients
# of the step's shapes 8.67 ms at 768, 9.72 at 384, 12.97 at 256): in the step it shares the GPU with the main stream's
# kernels and the package power cap with everything, and what counts there is bytes and co-residency, not
# its own critical path.  Earlier rounds (one chunk per sample, other kernels): 768 -> 4112 meshes/s, 1536 -> 4066, 2560 -> 4008,
# 4096 -> 3698; round 6: 1536 blocks (three full rounds of the slots) instead of 768 at the finest level 8.44 vs 8.45-8.52 ms.
# Later the same day, two more boxes: 384 -> 41.60-41.65, 256 -> 41.72, 224 / 208 -> 41.8-41.96 (3 samples per chunk: 86 chunks,
# the last one short), **192 -> 40.98-41.03** (4 samples per chunk at the finest levels, 8 / 16 below: 192 blocks everywhere),
# 176 -> 41.16, 160 -> 41.40, 150 -> 41.8, 128 -> 42.6; with the kernel held to ONE block per CU (16 KB more LDS): 192 -> 41.3-41.4,
# 128 / 96 -> 42.6-42.9 (`profiles/r06_tn_target_blocks_sweep.txt`).
TN_TARGET_BLOCKS = int(_os.environ.get("P2M_TN_TARGET_BLOCKS", "192"))
TN_SAMPLES_PER_CHUNK = _os.environ.get("P2M_TN_SPC", "1") == "1"       # 0: one chunk per sample (or less), the A/B form


def gemm_tn_rows(g, row_set, B, A, Ka, a0_shift, G, Gc, compact, a_amax=None, g_amax=None, g_bits=0, a_act=None,
                 samples_per_chunk=None):
    """Weight-gradient partials over a row set: returns (P[nchunks, Ka, len(G)*Gc], Pdb, nchunks).  f16x2: the amax
    words of A and of the G planes (after g_bits binades); default: the tensors' own.  a_act = (scale[Ka], shift[Ka]):
    activation on load of A (a raw conv output); a_amax must then bound the activated operand.  samples_per_chunk: whole
    samples per chunk (>= 2, slice arithmetics; default: the policy at TN_SAMPLES_PER_CHUNK)."""
    n = g.set_size(row_set)
    if a_act is not None and a_amax is None and f16x2():
        raise P2MError("gemm_tn_rows: activation on load needs the amax word of the activated operand (act_bound)")
    if f16x2():
        if a_amax is None:
            a_amax = amax_of(A, g if (a0_shift == 0 and row_set <= 2) else None, B)
        if g_amax is None:
            g_amax = _amax_planes(G, g if row_set <= 2 else None, B)
    N = len(G) * Gc
    ntiles = ((Ka + 127) // 128) * ((N + 127) // 128)
    splits = max(1, -(-TN_TARGET_BLOCKS // (B * ntiles)))
    nch = B * splits
    spc = (B * ntiles) // TN_TARGET_BLOCKS if (TN_SAMPLES_PER_CHUNK and GEMM_ARITH != "f32") else 1
    if samples_per_chunk is not None:
        spc = int(samples_per_chunk)
    if spc >= 2:                    # many tiles, few rows per sample: several whole samples per chunk (include/p2m.h)
        splits, nch = -spc, -(-B // spc)
    P = torch.empty((nch, Ka, N), device=A.device, dtype=torch.float32)
    Pdb = torch.empty((nch, N), device=A.device, dtype=torch.float32)
    gp = [_p(_req(t, "G plane")) for t in G] + [None] * (3 - len(G))
    # algorithmic HBM bytes: every row of A and of the G planes once (the partials are small)
    with _timed("gemm_tn_mfma", (2.0 * B * n * Ka * N, 2.0 * B * n * Ka * N, 4.0 * B * n * (Ka + N))):
        check(_lib.hip().p2m_gemm_tn_rows(g.handle, row_set, B, _p(_req(A, "A")), Ka, a0_shift, gp[0], gp[1], gp[2],
                                          len(G), Gc, int(compact), splits, _p(P), _p(Pdb), arith_code(),
                                          _p(a_amax), _p(g_amax), int(g_bits),
                                          _p(None if a_act is None else _req(a_act[0], "a_scale")),
                                          _p(None if a_act is None else _req(a_act[1], "a_shift")), _stream()),
              "p2m_gemm_tn_rows")
    return P, Pdb, nch


def weight_eff(Wt, Ka, N, a, b):
    We = torch.empty((Ka, N), device=Wt.device, dtype=torch.float32)
    check(_lib.hip().p2m_weight_eff(_p(_req(Wt, "Wt")), _p(We), Ka, N, float(a), float(b), _stream()), "p2m_weight_eff")
    return We


def weight_grad_unpack2(P, Pdb, nch, P2, Pdb2, nch2, s1, s2, Fout, Fin, dW=None, db=None):
    """dW/db given: ACCUMULATE into them (the parameter's .grad, e.g. a slice of the flat gradient buffer)."""
    acc = 1 if dW is not None else 0
    if dW is None:
        dW = torch.empty((Fout, Fin * 3), device=P.device, dtype=torch.float32)
        db = torch.empty((Fout,), device=P.device, dtype=torch.float32)
    check(_lib.hip().p2m_weight_grad_unpack2(_p(P), _p(Pdb), nch, _p(P2), _p(Pdb2), nch2, float(s1), float(s2), _p(dW),
                                             _p(db), Fout, Fin, 3, acc, _stream()), "p2m_weight_grad_unpack2")
    return dW, db


def bn_finalize_rows(g, B, st_real, st_fake, gamma, beta, running_mean, running_var, momentum, eps):
    N = gamma.shape[0]
    co = torch.empty((4, N), device=gamma.device, dtype=torch.float32)
    check(_lib.hip().p2m_bn_finalize_split(g.handle, _p(st_real), _p(st_fake), B, _p(_req(gamma, "bn.weight")),
                                           _p(_req(beta, "bn.bias")), _p(running_mean), _p(running_var),
                                           float(momentum), float(eps), _p(co[0]), _p(co[1]), _p(co[2]), _p(co[3]), N,
                                           _stream()), "p2m_bn_finalize_split")
    return co


def weight_pack(W, Fin, K, need_w2=True, need_w3=False):
    """Wt [k*Fin+fin][Fout], W2 [Fout][k*Fin+fin] (optional), W3 [k*Fout+fout][Fin] (optional)."""
    Fout = W.shape[0]
    Wt = torch.empty((K * Fin, Fout), device=W.device, dtype=torch.float32)
    W2 = torch.empty((Fout, K * Fin), device=W.device, dtype=torch.float32) if need_w2 else None
    W3 = torch.empty((K * Fout, Fin), device=W.device, dtype=torch.float32) if need_w3 else None
    check(_lib.hip().p2m_weight_pack(_p(_req(W, "weight")), _p(Wt), _p(W2), _p(W3), Fout, Fin, K, _stream()),
          "p2m_weight_pack")
    return Wt, W2, W3


def gemm_planes(A, Ka, a0_shift, Bm, bias, M, N, nplanesC=1, stats=False, addend=None, pair_out=False, Bx=None,
                act=None, amax=None, amax_bits=0, want_amax=False):
    """A: list of 1..3 plane tensors.  Returns (list of C planes, stats or None).
    addend: [M, N] added in the epilogue; pair_out: the (single) output has M/2 rows = sums of row pairs.
    Bx: cached weight_split(Bm), if the caller has one.  act = (scale[N], shift[N], relu): fused eval-mode
    BatchNorm + ReLU in the epilogue (bitwise the separate bn_act_fwd pass).  f16x2: amax = the word bounding the A
    planes after amax_bits binades (default: A[0]'s own, over the whole tensor); want_amax: the (single) output comes
    back tagged with its bound."""
    dev = A[0].device
    Nc = N // nplanesC
    C = [torch.empty((M >> 1 if pair_out else M, Nc), device=dev, dtype=torch.float32) for _ in range(nplanesC)]
    st = None
    if stats:
        nt = (M + stats_tile_rows() - 1) // stats_tile_rows()
        st = torch.empty((nt, 2, N), device=dev, dtype=torch.float32)
    a = [_p(_req(t, "A plane")) for t in A] + [None] * (3 - len(A))
    c = [_p(t) for t in C] + [None] * (3 - len(C))
    mfma = (Ka % 32 == 0) and (Nc % 32 == 0)
    if not mfma:
        Bx = None
    elif Bx is None:
        Bx = weight_split(Bm)
    amax_out = None
    if mfma and f16x2():
        if amax is None:
            amax = _amax_planes(A)
        if want_amax and nplanesC == 1:
            amax_out = new_amax(dev)
            tag_amax(C[0], amax_out)
    fl = 2.0 * M * len(A) * Ka * N                                                                  # algorithmic FLOPs
    nbytes = 4.0 * M * ((len(A) - 1 + 1.0 / (1 << a0_shift)) * Ka + (0.5 if pair_out else 1.0) * N
                        + (N if addend is not None else 0))
    with _timed("gemm_planes_mfma" if mfma else "gemm_planes_valu", (fl, fl, nbytes)):
        check(_lib.hip().p2m_gemm_planes(a[0], a[1], a[2], len(A), Ka, a0_shift, _p(_req(Bm, "B")), _p(Bx),
                                         arith_code() if mfma else 0, _p(amax), int(amax_bits),
                                         _p(bias if bias is None else _req(bias, "bias")),
                                         _p(addend if addend is None else _req(addend, "addend")), c[0], c[1], c[2],
                                         nplanesC, Nc, int(pair_out), M, _p(st),
                                         _p(None if act is None else act[0]), _p(None if act is None else act[1]),
                                         int(bool(act and act[2])), _p(amax_out), _stream()), "p2m_gemm_planes")
    return C, st


def pick_chunk_rows(M, ntiles_out, target_blocks=None, quantum=32):
    """Rows per split for the weight-gradient contraction: enough blocks to fill 256 CUs."""
    if target_blocks is None:
        target_blocks = TN_TARGET_BLOCKS
    nchunks = max(1, min((target_blocks + ntiles_out - 1) // ntiles_out, (M + 255) // 256))
    rows = (M + nchunks - 1) // nchunks
    rows = ((rows + quantum - 1) // quantum) * quantum
    return rows


def gemm_tn(A, Ka, a0_shift, G, M, N, a_amax=None, a_bits=0, g_amax=None, g_bits=0, tname=None, target_blocks=None):
    """G: one [M, N] tensor or a list of column planes [M, N/len(G)].
    Returns (P[nchunks, len(A)*Ka, N], Pdb[nchunks, N], nchunks).  f16x2: the amax words of the A / G planes (after
    a_bits / g_bits binades); default: the first planes' own, over the whole tensors.  tname: label of the launch in
    the per-kernel timing (default gemm_tn_mfma / gemm_tn_valu: the weight-gradient family of bench.py's roofline);
    target_blocks: blocks the launch should have at least (default TN_TARGET_BLOCKS)."""
    Gl = list(G) if isinstance(G, (list, tuple)) else [G]
    G = Gl[0]
    Gc = N // len(Gl)
    Ktot = len(A) * Ka
    ntiles = ((Ktot + 127) // 128) * ((N + 127) // 128)
    mfma = (Ka % 4 == 0) and (N % 32 == 0) and Ktot >= 32 and Gc % 4 == 0
    # the scalar fall-back (the first conv's 15 x 32 gradient) walks a chunk's rows one dependent load at a time: short
    # chunks (32 rows: 94 -> 15 us per step for the 4 864 joint rows of a batch of 256), the unpack sums them
    chunk_rows = pick_chunk_rows(M, ntiles, target_blocks) if mfma else 32
    nchunks = (M + chunk_rows - 1) // chunk_rows
    P = torch.empty((nchunks, Ktot, N), device=G.device, dtype=torch.float32)
    Pdb = torch.empty((nchunks, N), device=G.device, dtype=torch.float32)
    a = [_p(_req(t, "A plane")) for t in A] + [None] * (3 - len(A))
    gp = [_p(_req(t, "G plane")) for t in Gl] + [None] * (3 - len(Gl))
    if mfma and f16x2():
        if a_amax is None:
            a_amax = _amax_planes(A)
        if g_amax is None:
            g_amax = _amax_planes(Gl)
    with _timed(tname or ("gemm_tn_mfma" if mfma else "gemm_tn_valu"),
                (2.0 * M * Ktot * N, 2.0 * M * Ktot * N, 4.0 * M * (Ktot + N))):
        check(_lib.hip().p2m_gemm_tn(a[0], a[1], a[2], len(A), Ka, a0_shift, gp[0], gp[1], gp[2], len(Gl), Gc, M,
                                     chunk_rows, _p(P), _p(Pdb), arith_code() if mfma else 0, _p(a_amax), int(a_bits),
                                     _p(g_amax), int(g_bits), _stream()), "p2m_gemm_tn")
    return P, Pdb, nchunks


def weight_grad_unpack(P, Pdb, nchunks, Fout, Fin, K, dW=None, db=None, layout=0):
    """layout 0: P[c][k*Fin+fin][fout]; layout 1: P[c][fin][k*Fout+fout]; layout 2 (K = 1): P[c][fout][fin].  Output in
    nn.Linear layout.  Pdb = None: no bias gradient (returned / left as it is)."""
    acc = 1 if dW is not None else 0
    if dW is None:
        dW = torch.empty((Fout, Fin * K), device=P.device, dtype=torch.float32)
        db = torch.empty((Fout,), device=P.device, dtype=torch.float32) if Pdb is not None else None
    check(_lib.hip().p2m_weight_grad_unpack(_p(P), _p(Pdb), nchunks, _p(dW), _p(db if Pdb is not None else None), Fout, Fin,
                                            K, acc, layout, Pdb.shape[1] if Pdb is not None else 0, _stream()),
          "p2m_weight_grad_unpack")
    return dW, db


def bn_finalize(stats, M, gamma, beta, running_mean, running_var, momentum, eps, tile_rows=None):
    """stats: per-tile partials [>= ceil(M/tile_rows)][2][N] from p2m_gemm_planes (128-row tiles)."""
    N = gamma.shape[0]
    if tile_rows is None:
        tile_rows = stats_tile_rows()
    ntiles = (M + tile_rows - 1) // tile_rows
    co = torch.empty((4, N), device=gamma.device, dtype=torch.float32)  # mean, invstd, scale, shift
    check(_lib.hip().p2m_bn_finalize(_p(stats), ntiles, M, _p(_req(gamma, "bn.weight")),
                                     _p(_req(beta, "bn.bias")), _p(running_mean), _p(running_var),
                                     float(momentum), float(eps), _p(co[0]), _p(co[1]), _p(co[2]), _p(co[3]), N,
                                     tile_rows, _stream()), "p2m_bn_finalize")
    return co


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    N = gamma.shape[0]
    co = torch.empty((4, N), device=gamma.device, dtype=torch.float32)
    check(_lib.hip().p2m_bn_eval_coeffs(_p(_req(gamma, "bn.weight")), _p(_req(beta, "bn.bias")),
                                        _p(_req(running_mean, "running_mean")), _p(_req(running_var, "running_var")),
                                        float(eps), _p(co[0]), _p(co[1]), _p(co[2]), _p(co[3]), N, _stream()),
          "p2m_bn_eval_coeffs")
    return co


def bn_act_fwd(y, co, relu, resid, Fres, res_shift, M, F, classes=None, real_rows=None):
    """classes: the level's DeviceGraph when the holes of y hold no data (they are then skipped); real_rows: the level's
    DeviceGraph when only the REAL rows of y hold data (inference on the real rows): only those are walked.  f16x2: the
    output comes back tagged with the amax word of what was written."""
    x = torch.empty((M, F), device=y.device, dtype=torch.float32)
    sc = None if co is None else co[2]
    sh = None if co is None else co[3]
    cls = classes.handle if (classes is not None and classes.classes) else None
    if real_rows is not None:
        cls = real_rows.handle
    mode = 1 if real_rows is not None else 0
    word = new_amax(y.device) if (f16x2() and F % 4 == 0) else None
    check(_lib.hip().p2m_bn_act_fwd(_p(_req(y, "y")), _p(sc), _p(sh), int(relu),
                                    _p(resid if resid is None else _req(resid, "resid")), int(Fres), int(res_shift),
                                    _p(x), M, F, cls, mode, _p(word), _stream()), "p2m_bn_act_fwd")
    if word is not None:
        tag_amax(x, word)
    return x


def bnr_parts(g, plan, N, B, device):
    """The partial-sum array of a BatchNorm-backward reduction split between the tile kernel (real-vertex rows, slots
    [nfake:]) and p2m_bn_bwd_reduce_fake (slots [:nfake]), or None when the fused form is not available."""
    slots = tile_bnr_slots(g, plan, N, B)
    if slots <= 0:
        return None
    nfake = int(_lib.hip().p2m_bn_bwd_blocks_fake(g.handle, B, N)) if g.n_fake > 0 else 0
    part = torch.empty((nfake + slots, 2, N), device=device, dtype=torch.float32)
    return part, nfake


def bn_relu_bwd(gx, y, co, gamma, relu, training, M, F, dgamma=None, dbeta=None, pair_in=False, pair_out=False,
                classes=None, zero_holes=False, real_part=None):
    """Returns (gy, dgamma, dbeta).  dgamma/dbeta given: ACCUMULATE into them (the parameters' .grad).
    pair_in / pair_out: also return the pair-sums [M/2, F] of gx / of gy as by-products of the apply pass:
    (gy, dgamma, dbeta, pair_gx or None, pair_gy or None).
    classes: the level's DeviceGraph (include/p2m.h "classes"): holes are skipped -- not read, not written; zero_holes:
    the outputs are zero there instead of undefined (levels whose other kernels walk ALL rows).
    real_part = (g, part [nfake + slots, 2, F], nfake): the real-vertex rows of the reduction were already summed into
    part[nfake:] by the kernel that produced gx (cheb_tile_gemm bnr=); only the fake-vertex rows are reduced here."""
    lib = _lib.hip()
    cls = classes.handle if (classes is not None and classes.classes) else None
    if real_part is not None:
        gfr, part, nfake = real_part
        nblk = part.shape[0]
    else:
        nblk = int(lib.p2m_bn_bwd_blocks(M, F) if cls is None else lib.p2m_bn_bwd_blocks_classes(cls, M, F))
        part = torch.empty((nblk, 2, F), device=y.device, dtype=torch.float32)
    acc = 1 if dgamma is not None else 0
    if dgamma is None:
        dgb = torch.empty((2, F), device=y.device, dtype=torch.float32)
        dgamma, dbeta = dgb[0], dgb[1]
    coef = torch.empty((2, F), device=y.device, dtype=torch.float32)
    if real_part is not None:
        if nfake > 0:
            check(lib.p2m_bn_bwd_reduce_fake(gfr.handle, _p(_req(gx, "gx")), _p(_req(y, "y")), _p(co[2]), _p(co[3]), _p(co[0]),
                                             _p(co[1]), int(relu), _p(part), M // gfr.V, F, _stream()),
                  "p2m_bn_bwd_reduce_fake")
    else:
        check(lib.p2m_bn_bwd_reduce(_p(_req(gx, "gx")), _p(_req(y, "y")), _p(co[2]), _p(co[3]), _p(co[0]), _p(co[1]),
                                    int(relu), _p(part), M, F, cls, _stream()), "p2m_bn_bwd_reduce")
    check(lib.p2m_bn_bwd_finalize(_p(part), nblk, M, _p(dgamma), _p(dbeta), _p(coef), acc, F, _stream()),
          "p2m_bn_bwd_finalize")
    gy = torch.empty((M, F), device=y.device, dtype=torch.float32)      # (zero_holes: the pass itself stores the zeros)
    word = new_amax(y.device) if f16x2() else None
    pgx = torch.empty((M // 2, F), device=y.device, dtype=torch.float32) if pair_in else None
    pgy = torch.empty((M // 2, F), device=y.device, dtype=torch.float32) if pair_out else None
    check(lib.p2m_bn_bwd_apply(_p(gx), _p(y), _p(co[2]), _p(co[3]), _p(co[0]), _p(co[1]), _p(_req(gamma, "bn.weight")),
                               _p(coef) if training else None, int(relu), _p(gy), _p(pgx), _p(pgy), M, F, cls,
                               int(bool(zero_holes and cls is not None)), _p(word), _stream()), "p2m_bn_bwd_apply")
    if word is not None:
        tag_amax(gy, word)
    if pair_in or pair_out:
        return gy, dgamma, dbeta, pgx, pgy
    return gy, dgamma, dbeta


def pair_sum(x, Mout, F, classes=None):
    """classes: the DeviceGraph of x's level when it has classes (holes are left out of the sums)."""
    out = torch.empty((Mout, F), device=x.device, dtype=torch.float32)
    cls = classes.handle if (classes is not None and classes.classes) else None
    check(_lib.hip().p2m_pair_sum(_p(_req(x, "in")), _p(out), Mout, F, cls, _stream()), "p2m_pair_sum")
    return out


def class_reduce(g, G, B, F):
    """Gradient in class-sum form: representatives get the sum over their class, holes 0 (include/p2m.h "classes")."""
    out = torch.empty_like(G)
    check(_lib.hip().p2m_class_reduce(g.handle, _p(_req(G, "G")), _p(out), B, F, _stream()), "p2m_class_reduce")
    return out


def lerp_bwd_add(g, dst, M, F, Fres):
    check(_lib.hip().p2m_lerp_bwd_add(_p(_req(g, "g")), _p(_req(dst, "dst")), M, F, Fres, _stream()),
          "p2m_lerp_bwd_add")
    return dst


# ---- PoseNet (lib/models/posenet.py): the elementwise stage between two weight-streaming contractions ----------------
def gemm_tn_acc(A, G, P, a_amax=None, g_amax=None):
    """P[k][n] += sum_r A[r][k] * G[r][n] (A: [M, Ka], G: [M, N], P: [Ka, N] fp32, e.g. a parameter's .grad): the whole
    reduction in one chunk, no partial buffer (include/p2m.h p2m_gemm_tn_acc)."""
    M, Ka = A.shape
    N = G.shape[1]
    if G.shape[0] != M or tuple(P.shape) != (Ka, N):
        raise P2MError(f"gemm_tn_acc: shapes {tuple(A.shape)} x {tuple(G.shape)} -> {tuple(P.shape)} do not match")
    mfma = Ka % 4 == 0 and N % 32 == 0 and Ka >= 32
    if mfma and f16x2():
        if a_amax is None:
            a_amax = amax_of(A)
        if g_amax is None:
            g_amax = amax_of(G)
    with _timed("pn_gemm", (2.0 * M * Ka * N, 2.0 * M * Ka * N, 4.0 * (M * (Ka + N) + 2.0 * Ka * N))):
        check(_lib.hip().p2m_gemm_tn_acc(_p(_req(A, "A")), Ka, _p(_req(G, "G")), N, M, _p(_req(P, "P")),
                                         arith_code() if mfma else 0, _p(a_amax), _p(g_amax), _stream()), "p2m_gemm_tn_acc")
    return P


def pn_stage_fwd(P, nch, B, F, bias=None, resid=None, want_z=True, bn=None, rnd=None, p_drop=0.0, want_a=True,
                 want_aT=True, B_real=0):
    """z = sum of the nch partials P[ch] + bias (+ resid); bn = (gamma, beta, running_mean, running_var, momentum, eps,
    training) -> a = dropout(relu(batch_norm(z))), else a = z; a row-major and / or transposed ([F, B]).  Returns
    (z, a, aT, mean, invstd); a / aT come back tagged with their amax word in f16x2 mode (include/p2m.h p2m_pn_stage_fwd).
    B_real: rows that hold samples when the batch is zero-padded (0: all B)."""
    dev = P.device
    z = torch.empty((B, F), device=dev, dtype=torch.float32) if want_z else None
    a = torch.empty((B, F), device=dev, dtype=torch.float32) if want_a else None
    aT = torch.empty((F, B), device=dev, dtype=torch.float32) if want_aT else None
    mean = invstd = None
    gamma = beta = rm = rv = None
    mom, eps, training = 0.1, 1e-5, 0
    if bn is not None:
        gamma, beta, rm, rv, mom, eps, training = bn
        st = torch.empty((2, F), device=dev, dtype=torch.float32)
        mean, invstd = st[0], st[1]
    word = new_amax(dev) if (f16x2() and (want_a or want_aT)) else None
    check(_lib.hip().p2m_pn_stage_fwd(_p(_req(P, "P")), int(nch), _p(bias), _p(resid), _p(z), int(bn is not None),
                                      int(bool(training)), _p(gamma), _p(beta), _p(rm), _p(rv), float(mom), float(eps),
                                      _p(rnd), float(p_drop), _p(a), _p(aT), _p(mean), _p(invstd), _p(word), B, F,
                                      int(B_real), _stream()), "p2m_pn_stage_fwd")
    if word is not None:
        for t in (a, aT):
            if t is not None:
                tag_amax(t, word)
    return z, a, aT, mean, invstd


def pn_stage_bwd(P, nch, B, F, addend=None, bn=None, rnd=None, p_drop=0.0, want_T=True, dgamma=None, dbeta=None,
                 dbias=None, accumulate=False, B_real=0):
    """Backward of pn_stage_fwd.  bn = (z, mean, invstd, gamma, beta, training).  Returns (gz, gzT); dgamma / dbeta / dbias
    (given tensors) are overwritten, or added into when accumulate (include/p2m.h p2m_pn_stage_bwd)."""
    dev = P.device
    gz = torch.empty((B, F), device=dev, dtype=torch.float32)
    gzT = torch.empty((F, B), device=dev, dtype=torch.float32) if want_T else None
    z = mean = invstd = gamma = beta = None
    training = 0
    if bn is not None:
        z, mean, invstd, gamma, beta, training = bn
    word = new_amax(dev) if f16x2() else None
    check(_lib.hip().p2m_pn_stage_bwd(_p(_req(P, "P")), int(nch), _p(addend), int(bn is not None), int(bool(training)),
                                      _p(z), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(rnd), float(p_drop), _p(gz),
                                      _p(gzT), _p(dgamma), _p(dbeta), _p(dbias), int(bool(accumulate)), _p(word), B, F,
                                      int(B_real), _stream()), "p2m_pn_stage_bwd")
    if word is not None:
        tag_amax(gz, word)
        if gzT is not None:
            tag_amax(gzT, word)
    return gz, gzT

"""taxator_tk_amd -- Python face of the MI355X-native RPA engine (ctypes over the C ABI in include/taxator_rpa.h).

The directory is named `taxator-tk_amd`; import it through `__graft_entry__.load_package()` (module name
`taxator_tk_amd`).  Everything here is plumbing around `librpa_hip.so` (hand-written HIP for gfx950, built by
`taxator-tk_amd/csrc/Makefile`).  There is NO CPU implementation behind this module: if the shared library is
missing or no gfx950 device answers, construction raises.  The CPU oracle lives in /oracle and is never imported here.

Mirrors the reference's seam (core/src/taxonpredictionmodelsequence.hh:326-339): `RPAPredictionModel(tax, query_store,
db_store, exclude_factor, reeval_bandwidth)` with `predict(batch)` for a whole batch of record sets.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RPA_LIB_PATH") or os.path.join(_HERE, "librpa_hip.so")  # RPA_LIB_PATH: A/B builds of the engine

DNA, PROTEIN = 0, 1
STORE_QUERY, STORE_DB = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
PROFILE_TIMING = 2  # RPAPredictionModel.profile(enable=PROFILE_TIMING): launch timing without the in-kernel counters
FLAG_IVAL_UNTOUCHED, FLAG_SHORTCUT, FLAG_SINGLE = 1, 2, 4

PRED_DTYPE = np.dtype([("lower", "<u4"), ("upper", "<u4"), ("rtax", "<u4"), ("support", "<u4"), ("begin", "<u4"),
                       ("end", "<u4"), ("ival", "<f4"), ("flags", "<u4")])
STATS_DTYPE = np.dtype([("n", "<u4"), ("pass0", "<u4"), ("pass1", "<u4"), ("pass2", "<u4")])
TRACE_DTYPE = np.dtype([("set", "<u4"), ("seq", "<u4"), ("pass", "<i4"), ("i", "<i4"), ("j", "<i4"), ("dist", "<f4"), ("sim", "<f4"),
                        ("reserved", "<u4")])
WINDOW_DTYPE = np.dtype([("seq_index", "<u4"), ("reserved", "<u4"), ("first", "<u8"), ("n", "<u8")])
ALN_DTYPE = np.dtype([("matches", "<u4"), ("mismatches", "<u4"), ("gaps", "<u4"), ("length", "<u4"),
                      ("distance", "<f4"), ("similarity", "<f4")])

ERROR_NAMES = {0: "OK", 1: "INVALID", 2: "DEVICE", 3: "RANGE", 4: "SEQ_ID", 5: "ASSERT", 6: "UNSUPPORTED"}

# every symbol include/taxator_rpa.h declares (checked by the CPU-side ABI test)
ABI_SYMBOLS = ["rpa_abi_version", "rpa_default_params", "rpa_create", "rpa_destroy", "rpa_last_error", "rpa_set_stream",
               "rpa_store_load", "rpa_predict_batch", "rpa_align_pairs", "rpa_profile_enable", "rpa_profile_get",
               "rpa_profile_reset", "rpa_lca_default_params", "rpa_lca_predict_batch", "rpa_store_reserve", "rpa_store_append",
               "rpa_store_info_get", "rpa_submit_batch", "rpa_wait", "rpa_host_register",
               "rpa_host_unregister", "rpa_debug_where", "rpa_store_reserve_windows", "rpa_store_append_windows", "rpa_store_reset", "rpa_trace_begin", "rpa_trace_read"]

LCA_DUMMY, LCA_SIMPLE, LCA_MEGAN, LCA_NBEST = 0, 1, 2, 3  # taxator -a dummy | simple-lca | megan-lca | n-best-lca


class RpaError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("rpa error %d (%s): %s" % (code, ERROR_NAMES.get(code, "?"), text))
        self.code = code


class _Taxonomy(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("parent", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p),
                ("depth", C.c_void_p)]


class _Params(C.Structure):
    _fields_ = [("alphabet", C.c_int), ("exclude_alignments_factor", C.c_float), ("reeval_bandwidth", C.c_float),
                ("device", C.c_int), ("speculation_pass1", C.c_uint32), ("speculation_pass2", C.c_uint32),
                ("reserved", C.c_uint32 * 2)]


class _Batch(C.Structure):
    _fields_ = [("n_sets", C.c_uint32), ("n_records", C.c_uint32), ("set_offsets", C.c_void_p), ("query_seq", C.c_void_p),
                ("query_len", C.c_void_p), ("qstart", C.c_void_p), ("qstop", C.c_void_p), ("rstart", C.c_void_p),
                ("rstop", C.c_void_p), ("score", C.c_void_p), ("identities", C.c_void_p), ("alnlen", C.c_void_p),
                ("ref_seq", C.c_void_p), ("node", C.c_void_p), ("filtered", C.c_void_p), ("mem_kind", C.c_int)]


class _LcaParams(C.Structure):
    _fields_ = [("model", C.c_int), ("toppercent", C.c_float), ("minscore", C.c_float), ("maxevalue", C.c_double),
                ("minsupport", C.c_uint32), ("nbest", C.c_uint32), ("ignore_unclassified", C.c_int), ("reserved", C.c_uint32)]


class _Profile(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("pairs", C.c_uint64), ("residue_bytes", C.c_uint64), ("cells", C.c_uint64),
                ("align_ms", C.c_double), ("total_ms", C.c_double), ("demanded", C.c_uint64), ("rounds", C.c_uint64),
                ("cells_updated", C.c_uint64)]


_lib = None


def load_library(path=None):
    """dlopen librpa_hip.so and declare the ABI.  Raises if it was not built -- there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  If
    # torch is going to be used in this process (bench.py, torch.distributed) it has to be loaded FIRST so that this
    # library binds to the runtime that is already resident; two runtimes in one process fight over the device.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before HIP comes up: a hardware queue per engine stream (rpa_engine.hip: make_stream)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError("%s not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()')" % path)
    L = C.CDLL(path)
    L.rpa_abi_version.restype = C.c_int
    L.rpa_default_params.argtypes = [C.POINTER(_Params)]
    L.rpa_default_params.restype = None
    L.rpa_create.argtypes = [C.POINTER(_Taxonomy), C.POINTER(_Params), C.POINTER(C.c_void_p)]
    L.rpa_create.restype = C.c_int
    L.rpa_destroy.argtypes = [C.c_void_p]
    L.rpa_destroy.restype = None
    L.rpa_last_error.argtypes = [C.c_void_p]
    L.rpa_last_error.restype = C.c_char_p
    L.rpa_debug_where.argtypes = []
    L.rpa_debug_where.restype = C.c_char_p
    L.rpa_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.rpa_set_stream.restype = C.c_int
    L.rpa_store_load.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.rpa_store_load.restype = C.c_int
    L.rpa_predict_batch.argtypes = [C.c_void_p, C.POINTER(_Batch), C.c_void_p, C.c_void_p, C.c_int]
    L.rpa_predict_batch.restype = C.c_int
    L.rpa_align_pairs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int]
    L.rpa_align_pairs.restype = C.c_int
    L.rpa_store_reserve.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.rpa_store_reserve.restype = C.c_int
    L.rpa_store_append.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rpa_store_append.restype = C.c_int
    L.rpa_store_reserve_windows.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    L.rpa_store_reserve_windows.restype = C.c_int
    L.rpa_store_append_windows.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rpa_store_append_windows.restype = C.c_int
    L.rpa_store_reset.argtypes = [C.c_void_p, C.c_int]
    L.rpa_store_reset.restype = C.c_int
    L.rpa_trace_begin.argtypes = [C.c_void_p, C.c_uint64]
    L.rpa_trace_begin.restype = C.c_int
    L.rpa_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rpa_trace_read.restype = C.c_int
    L.rpa_store_info_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.rpa_store_info_get.restype = C.c_int
    L.rpa_submit_batch.argtypes = [C.c_void_p, C.POINTER(_Batch), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.rpa_submit_batch.restype = C.c_int
    L.rpa_wait.argtypes = [C.c_void_p, C.c_void_p]
    L.rpa_wait.restype = C.c_int
    L.rpa_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rpa_host_register.restype = C.c_int
    L.rpa_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
    L.rpa_host_unregister.restype = C.c_int
    L.rpa_lca_default_params.argtypes = [C.POINTER(_LcaParams)]
    L.rpa_lca_default_params.restype = None
    L.rpa_lca_predict_batch.argtypes = [C.c_void_p, C.POINTER(_Batch), C.c_void_p, C.c_void_p, C.POINTER(_LcaParams), C.c_void_p, C.c_int]
    L.rpa_lca_predict_batch.restype = C.c_int
    L.rpa_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.rpa_profile_enable.restype = C.c_int
    L.rpa_profile_get.argtypes = [C.c_void_p, C.POINTER(_Profile)]
    L.rpa_profile_get.restype = C.c_int
    L.rpa_profile_reset.argtypes = [C.c_void_p]
    L.rpa_profile_reset.restype = C.c_int
    if path == LIB_PATH:
        _lib = L
    return L


def _ptr(a):
    """numpy array -> host pointer; torch tensor / object with data_ptr() -> device (or host) pointer; int passes."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return int(a)


BATCH_FIELDS = (("set_offsets", np.uint32), ("query_seq", np.uint32), ("query_len", np.uint32), ("qstart", np.uint32),
                ("qstop", np.uint32), ("rstart", np.uint32), ("rstop", np.uint32), ("score", np.float32),
                ("identities", np.uint32), ("alnlen", np.uint32), ("ref_seq", np.uint32), ("node", np.uint32))


class _Job:
    """A batch in flight (rpa_submit_batch): holds the batch arrays and the output arrays until rpa_wait has returned."""

    def __init__(self, model, handle, rows, stats, keep):
        self._model, self._handle, self._rows, self._stats, self._keep = model, handle, rows, stats, keep
        self._done = False

    def wait(self):
        if not self._done:
            self._done = True  # the handle is released by rpa_wait whatever the status: never wait for it twice
            m, self._model = self._model, None
            try:
                m._check(m._L.rpa_wait(m._h, self._handle))
            finally:
                self._keep = None
        return self._rows, self._stats

    def __del__(self):
        try:
            if not self._done and self._model is not None and self._model._h and self._model._h.value:
                self.wait()
        except Exception:
            pass


class RPAPredictionModel:
    """One engine on one GPU.

    tax: object with .parent/.left/.right (uint32) and .depth (uint8) arrays, node 0 = root (pruned taxonomy).
    """

    def __init__(self, tax, alphabet=DNA, exclude_factor=0.5, reeval_bandwidth=0.05, device=0, speculation=None):
        self._L = load_library()
        self._h = C.c_void_p()
        parent = np.ascontiguousarray(tax.parent, np.uint32)
        left = np.ascontiguousarray(tax.left, np.uint32)
        right = np.ascontiguousarray(tax.right, np.uint32)
        depth = np.ascontiguousarray(tax.depth, np.uint8)
        t = _Taxonomy(len(parent), parent.ctypes.data, left.ctypes.data, right.ctypes.data, depth.ctypes.data)
        p = _Params()
        self._L.rpa_default_params(C.byref(p))
        p.alphabet, p.exclude_alignments_factor, p.reeval_bandwidth, p.device = alphabet, exclude_factor, reeval_bandwidth, device
        if speculation is not None:  # (pass-1 chunk, pass-2 chunk); 0 = whole set at once; results do not depend on it
            p.speculation_pass1, p.speculation_pass2 = speculation
        rc = self._L.rpa_create(C.byref(t), C.byref(p), C.byref(self._h))
        if rc:
            raise RpaError(rc, (self._L.rpa_last_error(None) or b"").decode())
        self.alphabet = alphabet
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rpa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise RpaError(rc, (self._L.rpa_last_error(self._h) or b"").decode())

    def set_stream(self, hip_stream):
        self._check(self._L.rpa_set_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else 0)))

    def load_store(self, which, letters, offsets, mem_kind=MEM_HOST):
        """letters: uint8 residue bytes back to back; offsets: uint64 [n_seqs+1] (numpy = host, torch cuda tensor = device)."""
        n = (len(offsets) if not hasattr(offsets, "numel") else offsets.numel()) - 1
        if isinstance(letters, np.ndarray):
            letters = np.ascontiguousarray(letters, np.uint8)
            offsets = np.ascontiguousarray(offsets, np.uint64)
        self._keep = (letters, offsets)
        self._check(self._L.rpa_store_load(self._h, which, n, _ptr(letters), _ptr(offsets), mem_kind))
        self._keep = None

    def pin(self, array):
        """Page-lock a numpy array that is handed to predict()/submit() again and again (rpa_host_register)."""
        self._check(self._L.rpa_host_register(self._h, array.ctypes.data, array.nbytes))

    def unpin(self, array):
        self._check(self._L.rpa_host_unregister(self._h, array.ctypes.data))

    def submit(self, batch, want_stats=True, out=None, stats=None, mem_kind=MEM_HOST):
        """Asynchronous predict: returns a job; job.wait() -> (rows, stats).  RPA_JOBS_IN_FLIGHT (4) jobs run at a time.
        mem_kind=MEM_DEVICE: batch / out / stats are torch cuda tensors that are complete (synchronise your stream first).
        The job object owns every buffer the engine reads or writes until wait() has returned; a job that is dropped without
        wait() (an exception between submit and wait) waits in its finaliser, so the buffers never go away under the engine."""
        if mem_kind == MEM_DEVICE:
            so, qs = batch["set_offsets"], batch["qstart"]
            b = _Batch(so.numel() - 1, qs.numel(), *[_ptr(batch[name]) for name, _ in BATCH_FIELDS], _ptr(batch.get("filtered")), MEM_DEVICE)
            handle = C.c_void_p()
            self._check(self._L.rpa_submit_batch(self._h, C.byref(b), _ptr(out), _ptr(stats), C.byref(handle)))
            return _Job(self, handle, out, stats, keep=(batch, b))
        arrs = {name: np.ascontiguousarray(batch[name], dt) for name, dt in BATCH_FIELDS}
        filt = batch.get("filtered")
        filt = None if filt is None else np.ascontiguousarray(filt, np.uint8)
        n_sets, n_rec = len(arrs["set_offsets"]) - 1, len(arrs["qstart"])
        b = _Batch(n_sets, n_rec, *[_ptr(arrs[name]) for name, _ in BATCH_FIELDS], _ptr(filt), MEM_HOST)
        rows = out if out is not None else np.zeros(n_sets, PRED_DTYPE)
        st = stats if stats is not None else (np.zeros(n_sets, STATS_DTYPE) if want_stats else None)
        handle = C.c_void_p()
        self._check(self._L.rpa_submit_batch(self._h, C.byref(b), rows.ctypes.data, None if st is None else st.ctypes.data, C.byref(handle)))
        return _Job(self, handle, rows, st, keep=(arrs, filt, b))

    def reserve_store(self, which, n_seqs):
        """An indexed store (sequencestorage.hh:318-406): n_seqs sequences known, none resident yet."""
        self._check(self._L.rpa_store_reserve(self._h, which, int(n_seqs)))

    def append_store(self, which, seq_index, letters, offsets):
        """Make the sequences seq_index[k] resident: letters back to back (uint8), offsets uint64 [n+1] from 0."""
        idx = np.ascontiguousarray(seq_index, np.uint32)
        letters = np.ascontiguousarray(letters, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self._check(self._L.rpa_store_append(self._h, which, len(idx), _ptr(idx), _ptr(letters), _ptr(offsets)))

    def trace(self, capacity):
        """Record the alignments the reference's control flow consumes in the following synchronous predict() calls
        (capacity entries per call; 0 = off)."""
        self._check(self._L.rpa_trace_begin(self._h, int(capacity)))
        self._trace_cap = int(capacity)

    def read_trace(self):
        """Entries of the last predict() call sorted by (set, seq) as a structured array, and the number that did not fit."""
        out = np.zeros(self._trace_cap, TRACE_DTYPE)
        n, dropped = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.rpa_trace_read(self._h, out.ctypes.data, self._trace_cap, C.byref(n), C.byref(dropped)))
        return out[: n.value], int(dropped.value)

    def reserve_windows(self, which, lengths):
        """A windowed store (region-granular residency): sequence i has lengths[i] residues, nothing resident yet."""
        lengths = np.ascontiguousarray(lengths, np.uint64)
        self._check(self._L.rpa_store_reserve_windows(self._h, which, len(lengths), _ptr(lengths)))

    def append_windows(self, which, seq_index, first, letters_list):
        """Make residues [first[k], first[k] + len(letters_list[k])) of sequence seq_index[k] readable (raw letters, uint8)."""
        n = len(seq_index)
        win = np.zeros(n, WINDOW_DTYPE)
        win["seq_index"] = seq_index
        win["first"] = first
        win["n"] = [len(x) for x in letters_list]
        offsets = np.zeros(n + 1, np.uint64)
        offsets[1:] = np.cumsum(win["n"])
        letters = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint8) for x in letters_list]) if n else np.zeros(0, np.uint8))
        self._check(self._L.rpa_store_append_windows(self._h, which, n, _ptr(win), _ptr(letters), _ptr(offsets)))

    def reset_store(self, which):
        self._check(self._L.rpa_store_reset(self._h, which))

    def store_info(self, which):
        class _Info(C.Structure):
            _fields_ = [("n_seqs", C.c_uint32), ("n_resident", C.c_uint32), ("residues", C.c_uint64), ("device_bytes", C.c_uint64),
                        ("windows", C.c_uint64), ("live_residues", C.c_uint64)]
        out = _Info()
        self._check(self._L.rpa_store_info_get(self._h, which, C.byref(out)))
        return {k: int(getattr(out, k)) for k, _ in _Info._fields_}

    def predict(self, batch, mem_kind=MEM_HOST, want_stats=True, out=None, stats=None):
        """batch: dict with the rpa_batch arrays (numpy for MEM_HOST, torch cuda tensors for MEM_DEVICE).
        Returns (rows, stats) as numpy structured arrays (host) or fills the given device tensors."""
        arrs = {}
        for name, dt in BATCH_FIELDS:
            a = batch[name]
            if mem_kind == MEM_HOST:
                a = np.ascontiguousarray(a, dt)
            arrs[name] = a
        filt = batch.get("filtered")
        if filt is not None and mem_kind == MEM_HOST:
            filt = np.ascontiguousarray(filt, np.uint8)
        so = arrs["set_offsets"]
        n_sets = (len(so) if isinstance(so, np.ndarray) else so.numel()) - 1
        qs = arrs["qstart"]
        n_rec = len(qs) if isinstance(qs, np.ndarray) else qs.numel()
        b = _Batch(n_sets, n_rec, *[_ptr(arrs[name]) for name, _ in BATCH_FIELDS], _ptr(filt), mem_kind)
        if mem_kind == MEM_HOST or out is None:
            rows = np.zeros(n_sets, PRED_DTYPE)
            st = np.zeros(n_sets, STATS_DTYPE) if want_stats else None
            self._check(self._L.rpa_predict_batch(self._h, C.byref(b), rows.ctypes.data, st.ctypes.data if want_stats else None, MEM_HOST))
            return rows, st
        self._check(self._L.rpa_predict_batch(self._h, C.byref(b), _ptr(out), _ptr(stats), MEM_DEVICE))
        return out, stats

    def lca_predict(self, batch, model, evalue=None, node_unclassified=None, toppercent=None, minscore=None, maxevalue=None,
                    minsupport=None, nbest=None, ignore_unclassified=False):
        """The reference's alignment-free models (taxator -a dummy|simple-lca|megan-lca|n-best-lca) on the device, for a
        host batch (numpy arrays; only set_offsets, query_len, qstart, qstop, score, node, filtered are read)."""
        need = {"set_offsets": np.uint32, "query_len": np.uint32, "qstart": np.uint32, "qstop": np.uint32, "score": np.float32,
                "node": np.uint32}
        arrs = {k: np.ascontiguousarray(batch[k], dt) for k, dt in need.items()}
        filt = batch.get("filtered")
        filt = None if filt is None else np.ascontiguousarray(filt, np.uint8)
        ev = None if evalue is None else np.ascontiguousarray(evalue, np.float64)
        un = None if node_unclassified is None else np.ascontiguousarray(node_unclassified, np.uint8)
        n_sets, n_rec = len(arrs["set_offsets"]) - 1, len(arrs["qstart"])
        ptrs = [_ptr(arrs[name]) if name in arrs else None for name, _ in BATCH_FIELDS]
        b = _Batch(n_sets, n_rec, *ptrs, _ptr(filt), MEM_HOST)
        p = _LcaParams()
        self._L.rpa_lca_default_params(C.byref(p))
        p.model = model
        for name, v in (("toppercent", toppercent), ("minscore", minscore), ("maxevalue", maxevalue), ("minsupport", minsupport),
                        ("nbest", nbest)):
            if v is not None:
                setattr(p, name, v)
        p.ignore_unclassified = 1 if ignore_unclassified else 0
        rows = np.zeros(n_sets, PRED_DTYPE)
        self._check(self._L.rpa_lca_predict_batch(self._h, C.byref(b), _ptr(ev), _ptr(un), C.byref(p), rows.ctypes.data, MEM_HOST))
        return rows

    def align_pairs(self, letters, offsets, pair_a, pair_b):
        """getAlignment(A, B) for explicit pairs of the sequences in (letters, offsets).  Host arrays in, numpy out."""
        letters = np.ascontiguousarray(letters, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        pair_a = np.ascontiguousarray(pair_a, np.uint32)
        pair_b = np.ascontiguousarray(pair_b, np.uint32)
        out = np.zeros(len(pair_a), ALN_DTYPE)
        self._check(self._L.rpa_align_pairs(self._h, len(offsets) - 1, letters.ctypes.data, offsets.ctypes.data, len(pair_a),
                                            pair_a.ctypes.data, pair_b.ctypes.data, out.ctypes.data, MEM_HOST))
        return out

    def profile(self, enable=None, reset=False):
        if enable is not None:
            self._check(self._L.rpa_profile_enable(self._h, PROFILE_TIMING if enable == PROFILE_TIMING and enable is not True else (1 if enable else 0)))
        if reset:
            self._check(self._L.rpa_profile_reset(self._h))
        p = _Profile()
        self._check(self._L.rpa_profile_get(self._h, C.byref(p)))
        return {f: getattr(p, f) for f, _ in _Profile._fields_}

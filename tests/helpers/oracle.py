"""ctypes bindings for the CHECKERS (test infrastructure): oracle/liboracle.so and oracle/_ref/libseqan_ref.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


class Taxonomy(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("parent", u32p), ("left", u32p), ("right", u32p), ("depth", u8p)]


class Store(C.Structure):
    _fields_ = [("residues", u8p), ("offsets", u64p), ("n_seqs", C.c_uint32)]


class Record(C.Structure):
    _fields_ = [("qstart", C.c_uint32), ("qstop", C.c_uint32), ("rstart", C.c_uint32), ("rstop", C.c_uint32),
                ("score", C.c_float), ("identities", C.c_uint32), ("alnlen", C.c_uint32), ("ref_seq", C.c_uint32),
                ("node", C.c_uint32), ("filtered", C.c_uint32)]


class LcaParams(C.Structure):  # orc_lca_params
    _fields_ = [("model", C.c_int), ("toppercent", C.c_float), ("minscore", C.c_float), ("maxevalue", C.c_double),
                ("minsupport", C.c_uint32), ("nbest", C.c_uint32), ("ignore_unclassified", C.c_int)]


LCA_DUMMY, LCA_SIMPLE, LCA_MEGAN, LCA_NBEST = 0, 1, 2, 3

RECORD_DTYPE = np.dtype([("qstart", "<u4"), ("qstop", "<u4"), ("rstart", "<u4"), ("rstop", "<u4"), ("score", "<f4"),
                         ("identities", "<u4"), ("alnlen", "<u4"), ("ref_seq", "<u4"), ("node", "<u4"),
                         ("filtered", "<u4")])
PRED_DTYPE = np.dtype([("lower", "<u4"), ("upper", "<u4"), ("rtax", "<u4"), ("support", "<u4"), ("begin", "<u4"),
                       ("end", "<u4"), ("ival", "<f4"), ("flags", "<u4")])
STATS_DTYPE = np.dtype([("n", "<u4"), ("pass0", "<u4"), ("pass1", "<u4"), ("pass2", "<u4")])
ALN_DTYPE = np.dtype([("matches", "<u4"), ("mismatches", "<u4"), ("gaps", "<u4"), ("length", "<u4"),
                      ("distance", "<f4"), ("similarity", "<f4"), ("selfscore", "<i4"), ("mutualscore", "<i4")])
assert C.sizeof(Record) == RECORD_DTYPE.itemsize

FLAG_IVAL_UNTOUCHED, FLAG_SHORTCUT, FLAG_SINGLE = 1, 2, 4
DNA, PROTEIN = 0, 1
TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float)

_lib = None
_ref = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_myers_score.restype = C.c_int
        L.orc_myers_score.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_levenshtein.restype = C.c_int
        L.orc_levenshtein.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_align_dna.restype = None
        L.orc_align_dna.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        for f in (L.orc_align_protein, L.orc_align_protein_forward):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_lca.restype = C.c_uint32
        L.orc_lca.argtypes = [C.POINTER(Taxonomy), C.c_uint32, C.c_uint32]
        L.orc_is_parent_of.restype = C.c_int
        L.orc_is_parent_of.argtypes = [C.POINTER(Taxonomy), C.c_uint32, C.c_uint32]
        L.orc_bandfactor.restype = C.c_float
        L.orc_bandfactor.argtypes = [C.POINTER(Taxonomy), C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_get_segment.restype = C.c_int
        L.orc_get_segment.argtypes = [C.POINTER(Store), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_void_p, u32p]
        L.orc_revcomp_dna5.restype = None
        L.orc_revcomp_dna5.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_sort_records.restype = None
        L.orc_sort_records.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_predict.restype = C.c_int
        L.orc_predict.argtypes = [C.POINTER(Taxonomy), C.POINTER(Store), C.POINTER(Store), C.c_int, C.c_float,
                                  C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
        L.orc_predict_batch.restype = C.c_int
        L.orc_predict_batch.argtypes = [C.POINTER(Taxonomy), C.POINTER(Store), C.POINTER(Store), C.c_int, C.c_float,
                                        C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.orc_lca_predict_batch.restype = C.c_int
        L.orc_lca_predict_batch.argtypes = [C.POINTER(Taxonomy), C.POINTER(LcaParams), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libseqan_ref.so"))


def ref():
    """The reference's own SeqAn, compiled in place (oracle/_ref). None if it was never built."""
    global _ref
    if _ref is None and ref_available():
        R = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libseqan_ref.so"))
        R.ref_align_dna.restype = None
        R.ref_align_dna.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        R.ref_align_protein.restype = None
        R.ref_align_protein.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_revcomp_dna5.restype = None
        R.ref_revcomp_dna5.argtypes = [C.c_void_p, C.c_int]
        _ref = R
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def as_u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def myers_score(a, b):
    a, b = as_u8(a), as_u8(b)
    return lib().orc_myers_score(_p(a), len(a), _p(b), len(b))


def levenshtein(a, b):
    a, b = as_u8(a), as_u8(b)
    return lib().orc_levenshtein(_p(a), len(a), _p(b), len(b))


def align_dna(a, b):
    a, b = as_u8(a), as_u8(b)
    out = np.zeros(1, ALN_DTYPE)
    lib().orc_align_dna(_p(a), len(a), _p(b), len(b), _p(out))
    return out[0]


def align_protein(a, b, forward=False):
    a, b = as_u8(a), as_u8(b)
    out = np.zeros(1, ALN_DTYPE)
    f = lib().orc_align_protein_forward if forward else lib().orc_align_protein
    rc = f(_p(a), len(a), _p(b), len(b), _p(out))
    return rc, out[0]


def ref_align_dna(a, b):
    a, b = as_u8(a), as_u8(b)
    out = np.zeros(5, np.int32)
    ref().ref_align_dna(_p(a), len(a), _p(b), len(b), _p(out))
    return out  # dist, match, mismatch, gap, length


def ref_align_protein(a, b):
    a, b = as_u8(a), as_u8(b)
    out = np.zeros(6, np.int32)
    f = np.zeros(2, np.float32)
    ref().ref_align_protein(_p(a), len(a), _p(b), len(b), _p(out), _p(f))
    return out, f  # selfscore, mutual, match, mismatch, gap, len ; distance, similarity


class TaxArrays:
    """Owns numpy arrays + the ctypes view of a flattened taxonomy."""

    def __init__(self, parent, left, right, depth):
        self.parent = np.ascontiguousarray(parent, np.uint32)
        self.left = np.ascontiguousarray(left, np.uint32)
        self.right = np.ascontiguousarray(right, np.uint32)
        self.depth = np.ascontiguousarray(depth, np.uint8)
        self.c = Taxonomy(len(self.parent), self.parent.ctypes.data_as(u32p), self.left.ctypes.data_as(u32p),
                          self.right.ctypes.data_as(u32p), self.depth.ctypes.data_as(u8p))

    def lca(self, a, b):
        return lib().orc_lca(C.byref(self.c), a, b)

    def is_parent_of(self, a, b):
        return bool(lib().orc_is_parent_of(C.byref(self.c), a, b))

    def bandfactor(self, dist, node):
        dist = np.ascontiguousarray(dist, np.float32)
        node = np.ascontiguousarray(node, np.uint32)
        return lib().orc_bandfactor(C.byref(self.c), _p(dist), _p(node), len(dist))


class StoreArrays:
    def __init__(self, residues, offsets):
        self.residues = as_u8(residues)
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        self.c = Store(self.residues.ctypes.data_as(u8p), self.offsets.ctypes.data_as(u64p), len(self.offsets) - 1)

    def segment(self, alphabet, seq, start, stop, left_ext=0, right_ext=0):
        cap = abs(int(stop) - int(start)) + 1 + left_ext + right_ext
        out = np.zeros(max(cap, 1), np.uint8)
        n = C.c_uint32(0)
        rc = lib().orc_get_segment(C.byref(self.c), alphabet, seq, start, stop, left_ext, right_ext, _p(out), C.byref(n))
        return rc, out[: n.value]


def predict(tax, qstore, dbstore, alphabet, query_seq, query_len, recs, exclude_factor=0.5, reeval_bandwidth=0.05,
            want_trace=False):
    recs = np.ascontiguousarray(recs, RECORD_DTYPE)
    out = np.zeros(1, PRED_DTYPE)
    st = np.zeros(1, STATS_DTYPE)
    trace = []
    cb = TRACE_FN(lambda ctx, p, i, j, d, s: trace.append((p, i, j, d, s))) if want_trace else None
    rc = lib().orc_predict(C.byref(tax.c), C.byref(qstore.c), C.byref(dbstore.c), alphabet, exclude_factor,
                           reeval_bandwidth, query_seq, query_len, _p(recs), len(recs), _p(out), _p(st),
                           C.cast(cb, C.c_void_p) if cb else None, None)
    return rc, out[0], st[0], trace


def predict_batch(tax, qstore, dbstore, alphabet, set_offsets, query_seq, query_len, recs, exclude_factor=0.5,
                  reeval_bandwidth=0.05):
    set_offsets = np.ascontiguousarray(set_offsets, np.uint32)
    query_seq = np.ascontiguousarray(query_seq, np.uint32)
    query_len = np.ascontiguousarray(query_len, np.uint32)
    recs = np.ascontiguousarray(recs, RECORD_DTYPE)
    n = len(set_offsets) - 1
    out = np.zeros(n, PRED_DTYPE)
    st = np.zeros(n, STATS_DTYPE)
    rc = lib().orc_predict_batch(C.byref(tax.c), C.byref(qstore.c), C.byref(dbstore.c), alphabet, exclude_factor,
                                 reeval_bandwidth, n, _p(set_offsets), _p(query_seq), _p(query_len), _p(recs), _p(out),
                                 _p(st))
    return rc, out, st


def lca_predict_batch(tax, model, set_offsets, query_len, recs, evalue=None, node_unclassified=None, toppercent=0.05, minscore=0.0,
                      maxevalue=1000.0, minsupport=1, nbest=1, ignore_unclassified=False):
    """The alignment-free models of the reference (oracle/lca_oracle.c) over a batch; returns (rc, rows)."""
    set_offsets = np.ascontiguousarray(set_offsets, np.uint32)
    query_len = np.ascontiguousarray(query_len, np.uint32)
    recs = np.ascontiguousarray(recs, RECORD_DTYPE)
    ev = None if evalue is None else np.ascontiguousarray(evalue, np.float64)
    un = None if node_unclassified is None else np.ascontiguousarray(node_unclassified, np.uint8)
    n = len(set_offsets) - 1
    out = np.zeros(n, PRED_DTYPE)
    p = LcaParams(model, toppercent, minscore, maxevalue, minsupport, nbest, 1 if ignore_unclassified else 0)
    rc = lib().orc_lca_predict_batch(C.byref(tax.c), C.byref(p), n, _p(set_offsets), _p(query_len), _p(recs),
                                     None if ev is None else _p(ev), None if un is None else _p(un), _p(out))
    return rc, out

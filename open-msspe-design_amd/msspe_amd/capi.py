"""ctypes binding of include/msspe_hip.h.  Fails loudly when the HIP library is missing."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent.parent          # open-msspe-design_amd/
STATUS = {0: "MSSPE_OK", 1: "MSSPE_ERR_ARG", 2: "MSSPE_ERR_K", 3: "MSSPE_ERR_TABLES",
          4: "MSSPE_ERR_DEVICE", 5: "MSSPE_ERR_CAPACITY", 6: "MSSPE_ERR_NOMEM"}

# every symbol include/msspe_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "msspe_chem_ntthal_defaults", "msspe_chem_primer3_defaults", "msspe_create", "msspe_destroy",
    "msspe_last_error", "msspe_version", "msspe_set_option", "msspe_get_info", "msspe_kmer_trace", "msspe_set_stream", "msspe_reset_stream",
    "msspe_synchronize",
    "msspe_pack_oligos", "msspe_unpack_oligo", "msspe_cross_dimer_dev", "msspe_cross_dimer",
    "msspe_cross_dimer_edges_dev", "msspe_cross_dimer_edges",
    "msspe_last_overflow_pairs", "msspe_pair_stage_stats", "msspe_pair_stage_samples", "msspe_host_pair_tables", "msspe_host_split_tables", "msspe_device_put_rows", "msspe_segment_coverage", "msspe_segment_coverage_dev",
    "msspe_device_put", "msspe_device_free", "msspe_thal_detail_pairs", "msspe_profile_enable", "msspe_profile_read",
    "msspe_oligo_stats_dev", "msspe_oligo_stats",
    "msspe_kmer_candidates", "msspe_kmer_candidates_dev", "msspe_round_g_f32",
    "msspe_packed_row_words", "msspe_device_put_rows_packed", "msspe_kmer_candidates_packed_dev",
    "msspe_kmer_candidates_both_packed_dev",
    "msspe_segment_coverage_packed_dev",
    "msspe_round_fixed_f32", "msspe_g_cut",
    "msspe_group_create", "msspe_group_destroy", "msspe_group_last_error", "msspe_group_size",
    "msspe_group_transport", "msspe_group_transport_reason", "msspe_group_rccl_available", "msspe_group_member", "msspe_group_set_option", "msspe_group_rows",
    "msspe_cross_dimer_group", "msspe_cross_dimer_edges_group", "msspe_oligo_stats_group",
]


class MsspeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class Chem(C.Structure):
    """msspe_chem: what od-msspe passes to ntthal (od-msspe/src/delta_g.rs:93-110)."""
    _fields_ = [("mv", C.c_double), ("dv", C.c_double), ("dntp", C.c_double),
                ("dna_conc", C.c_double), ("temp_c", C.c_double), ("max_loop", C.c_int)]

    @classmethod
    def ntthal(cls, mv=50.0, dv=3.0, dntp=0.0, dna_conc=250.0, temp_c=25.0, max_loop=30):
        return cls(mv, dv, dntp, dna_conc, temp_c, max_loop)

    @classmethod
    def primer3(cls):
        return cls(50.0, 1.5, 0.6, 50.0, 37.0, 30)


class KmerOpt(C.Structure):
    """msspe_kmer_opt: od-msspe/src/constants.rs:1-5 defaults."""
    _fields_ = [("segment_size", C.c_int), ("overlap_size", C.c_int),
                ("search_window_size", C.c_int), ("kmer_size", C.c_int),
                ("max_iterations", C.c_int), ("max_mismatch_segments", C.c_int)]


def lib_path() -> Path:
    return PKG_DIR / "libmsspe_hip.so"


_lib_override: Path | None = None


def use_library(path) -> None:
    """Development aid (A/B timing of two builds, tools/perf_probe.py MSSPE_PROBE_LIB): load another build of the library; call before the
    first Engine is created."""
    global _lib_override
    _lib_override = Path(path)


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    try:                      # PyTorch ships its own HIP runtime; when it is used in the same
        import torch          # process it must be loaded first so that a single runtime exists
    except Exception:         # (torch is plumbing here: device buffers, streams, torch.distributed)
        pass
    p = _lib_override or lib_path()
    if not p.exists():
        raise ImportError(f"{p} is missing: build it with open-msspe-design_amd/build.sh "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(str(p))
    vp, u64p = C.c_void_p, C.c_void_p
    L.msspe_version.restype = C.c_char_p
    L.msspe_last_error.restype = C.c_char_p
    L.msspe_last_error.argtypes = [vp]
    L.msspe_create.argtypes = [C.c_int, C.c_char_p, C.POINTER(vp)]
    L.msspe_destroy.argtypes = [vp]
    L.msspe_set_stream.argtypes = [vp, vp]
    L.msspe_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.msspe_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_longlong)]
    L.msspe_kmer_trace.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.msspe_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.msspe_group_destroy.argtypes = [vp]
    L.msspe_group_destroy.restype = None
    L.msspe_group_last_error.argtypes = [vp]
    L.msspe_group_last_error.restype = C.c_char_p
    L.msspe_group_size.argtypes = [vp]
    L.msspe_group_transport.argtypes = [vp]
    L.msspe_group_transport.restype = C.c_char_p
    L.msspe_group_transport_reason.argtypes = [vp]
    L.msspe_group_transport_reason.restype = C.c_char_p
    L.msspe_group_rccl_available.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.msspe_group_member.argtypes = [vp, C.c_int]
    L.msspe_group_member.restype = vp
    L.msspe_group_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.msspe_group_rows.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.msspe_cross_dimer_group.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Chem), C.c_float, vp, vp]
    L.msspe_cross_dimer_edges_group.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Chem), C.c_float, vp,
                                                C.c_uint64, C.POINTER(C.c_uint64)]
    L.msspe_oligo_stats_group.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Chem)] + [vp] * 5
    L.msspe_reset_stream.argtypes = [vp]
    L.msspe_synchronize.argtypes = [vp]
    L.msspe_pack_oligos.argtypes = [C.c_char_p, C.c_int, C.c_int, u64p]
    L.msspe_unpack_oligo.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
    L.msspe_cross_dimer_dev.argtypes = [vp, u64p, C.c_int, C.c_int, C.POINTER(Chem), C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.msspe_cross_dimer.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Chem), C.c_float,
                                    vp, vp, vp, vp]
    L.msspe_cross_dimer_edges.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Chem), C.c_float, vp,
                                          C.c_uint64, C.POINTER(C.c_uint64)]
    L.msspe_cross_dimer_edges_dev.argtypes = [vp, u64p, C.c_int, C.c_int, C.POINTER(Chem), C.c_float,
                                              C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_uint64, vp]
    L.msspe_last_overflow_pairs.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.msspe_pair_stage_stats.argtypes = [vp, C.POINTER(C.c_uint64)]   # out[16]
    L.msspe_pair_stage_samples.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]
    L.msspe_profile_enable.argtypes = [vp, C.c_int]
    L.msspe_profile_read.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    L.msspe_oligo_stats_dev.argtypes = [vp, u64p, C.c_int, C.c_int, C.POINTER(Chem)] + [vp] * 5
    L.msspe_oligo_stats.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(Chem)] + [vp] * 5
    L.msspe_kmer_candidates.argtypes = [vp, vp, C.c_int, C.c_size_t, C.POINTER(KmerOpt), C.c_int,
                                        vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.msspe_kmer_candidates_dev.argtypes = L.msspe_kmer_candidates.argtypes
    L.msspe_kmer_candidates_packed_dev.argtypes = L.msspe_kmer_candidates.argtypes
    L.msspe_kmer_candidates_both_packed_dev.argtypes = [vp, vp, C.c_int, C.c_size_t, C.POINTER(KmerOpt), vp, vp,
                                                        C.POINTER(C.c_int), vp, vp, C.POINTER(C.c_int), C.c_int]
    L.msspe_packed_row_words.restype = C.c_size_t
    L.msspe_packed_row_words.argtypes = [C.c_size_t]
    L.msspe_device_put_rows_packed.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_size_t,
                                               C.POINTER(vp)]
    L.msspe_segment_coverage.argtypes = [vp, vp, C.c_int, C.c_size_t, C.POINTER(KmerOpt), vp, C.c_int, vp, C.c_int,
                                         vp]
    L.msspe_segment_coverage_dev.argtypes = L.msspe_segment_coverage.argtypes
    L.msspe_segment_coverage_packed_dev.argtypes = L.msspe_segment_coverage.argtypes
    L.msspe_round_g_f32.restype = C.c_float
    L.msspe_round_g_f32.argtypes = [C.c_double]
    L.msspe_round_fixed_f32.restype = C.c_float
    L.msspe_round_fixed_f32.argtypes = [C.c_double, C.c_int]
    L.msspe_g_cut.restype = C.c_double
    L.msspe_g_cut.argtypes = [C.c_float]
    _lib = L
    return L


def pack_oligos(oligos) -> np.ndarray:
    """list[str] or uint8 (n,k) ASCII -> uint64[n] (2 bits per base, base p at bits 2p..2p+1)."""
    if isinstance(oligos, np.ndarray):
        n, k = oligos.shape
        buf = np.ascontiguousarray(oligos, dtype=np.uint8).tobytes()
    else:
        n = len(oligos)
        k = len(oligos[0]) if n else 0
        if any(len(o) != k for o in oligos):
            raise MsspeError(1, "oligos must all have the same length")
        buf = "".join(oligos).encode()
    out = np.zeros(n, dtype=np.uint64)
    rc = load_library().msspe_pack_oligos(buf, n, k, out.ctypes.data)
    if rc:
        raise MsspeError(rc, "pool holds characters other than ACGT" if rc == 1 else "bad oligo length")
    return out


def unpack_oligo(word: int, k: int) -> str:
    buf = C.create_string_buffer(k + 1)
    load_library().msspe_unpack_oligo(C.c_uint64(int(word)), k, buf)
    return buf.value.decode()


def round_g_f32(x: float) -> float:
    return float(load_library().msspe_round_g_f32(x))


def round_fixed_f32(x: float, decimals: int) -> float:
    return float(load_library().msspe_round_fixed_f32(x, decimals))


def g_cut(threshold: float) -> float:
    return float(load_library().msspe_g_cut(C.c_float(threshold)))


def _ascii(oligos):
    if isinstance(oligos, np.ndarray):
        n, k = oligos.shape
        return np.ascontiguousarray(oligos, dtype=np.uint8).tobytes(), n, k
    n = len(oligos)
    k = len(oligos[0]) if n else 0
    return "".join(oligos).encode(), n, k


class Engine:
    """One msspe_ctx bound to one device."""

    def __init__(self, device: int = 0, params_path: str | None = None):
        self.L = load_library()
        self.ptr = C.c_void_p()
        rc = self.L.msspe_create(device, params_path.encode() if params_path else None,
                                 C.byref(self.ptr))
        if rc:
            msg = self.L.msspe_last_error(self.ptr).decode() if self.ptr else "allocation failed"
            if self.ptr:
                self.L.msspe_destroy(self.ptr)
                self.ptr = C.c_void_p()
            raise MsspeError(rc, msg)
        self.device = device

    def close(self):
        if getattr(self, "ptr", None):
            self.L.msspe_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise MsspeError(rc, self.L.msspe_last_error(self.ptr).decode())

    def set_stream(self, hip_stream: int | None):
        """Run on the caller's HIP stream (raw handle; 0/None = HIP's default stream)."""
        self._check(self.L.msspe_set_stream(self.ptr, C.c_void_p(hip_stream or 0)))

    def set_option(self, key: str, value) -> None:
        """Engine option (include/msspe_hip.h msspe_set_option); the library never reads the environment."""
        self._check(self.L.msspe_set_option(self.ptr, key.encode(), str(value).encode()))

    def info(self, key: str) -> int:
        """Facts about the device and the kernels the context will run (include/msspe_hip.h msspe_get_info)."""
        v = C.c_longlong(0)
        self._check(self.L.msspe_get_info(self.ptr, key.encode(), C.byref(v)))
        return int(v.value)

    def kmer_trace(self) -> np.ndarray:
        """Per winner of the last kmer_candidates call: (iteration, how it was selected) -- msspe_kmer_trace."""
        n = C.c_int(0)
        self._check(self.L.msspe_kmer_trace(self.ptr, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        self._check(self.L.msspe_kmer_trace(self.ptr, out.ctypes.data, int(out.size), C.byref(n)))
        out = out[:n.value]
        return np.stack([out >> 8, out & 0xff], axis=1)

    def reset_stream(self):
        self._check(self.L.msspe_reset_stream(self.ptr))

    def synchronize(self):
        self._check(self.L.msspe_synchronize(self.ptr))

    # ---- stage C ---------------------------------------------------------------------------
    def cross_dimer(self, pool, chem: Chem | None = None, threshold: float = -9000.0,
                    want_dg=True, want_tm=False, want_bitmap=True):
        """Host-buffer call: returns dict(row_conflicts, bitmap, dg, tm) for the full n x n matrix."""
        buf, n, k = _ascii(pool)
        chem = chem or Chem.ntthal()
        words = (n + 63) // 64
        rc_ = np.zeros(n, dtype=np.uint32)
        bm = np.zeros((n, words), dtype=np.uint64) if want_bitmap else None
        dg = np.empty((n, n)) if want_dg else None
        tm = np.empty((n, n)) if want_tm else None
        self._check(self.L.msspe_cross_dimer(
            self.ptr, buf, n, k, C.byref(chem), C.c_float(threshold), rc_.ctypes.data,
            bm.ctypes.data if want_bitmap else None, dg.ctypes.data if want_dg else None,
            tm.ctypes.data if want_tm else None))
        return {"row_conflicts": rc_, "bitmap": bm, "dg": dg, "tm": tm}

    def cross_dimer_edges(self, pool, chem: Chem | None = None, threshold: float = -9000.0, capacity: int = 1 << 20):
        """Edge list of the whole pool: (edges structured array [a, b, dg], count); raises MsspeError
        (MSSPE_ERR_CAPACITY, .count = edges needed) when the capacity is too small."""
        buf, n, k = _ascii(pool)
        chem = chem or Chem.ntthal()
        edges = np.zeros(capacity, dtype=np.dtype([("a", np.uint32), ("b", np.uint32), ("dg", np.float32)]))
        count = C.c_uint64()
        rc = self.L.msspe_cross_dimer_edges(self.ptr, buf, n, k, C.byref(chem), C.c_float(threshold),
                                            edges.ctypes.data, capacity, C.byref(count))
        if rc:
            err = MsspeError(rc, self.L.msspe_last_error(self.ptr).decode())
            err.count = int(count.value)
            err.edges = edges
            raise err
        return edges[:count.value], int(count.value)

    def cross_dimer_edges_dev(self, d_pool: int, n: int, k: int, chem: Chem, threshold: float,
                              rows: tuple[int, int], cols: tuple[int, int], d_edges: int, capacity: int,
                              d_count: int, d_row_conflicts: int = 0):
        """Device-pointer edge list of a block (16-byte records a:u32, b:u32, dg:f64; *d_count may exceed
        the capacity = truncated); asynchronous."""
        self._check(self.L.msspe_cross_dimer_edges_dev(
            self.ptr, C.c_void_p(d_pool), n, k, C.byref(chem), C.c_float(threshold), rows[0], rows[1], cols[0],
            cols[1], C.c_void_p(d_row_conflicts), C.c_void_p(d_edges), capacity, C.c_void_p(d_count)))

    def cross_dimer_dev(self, d_pool: int, n: int, k: int, chem: Chem, threshold: float,
                        rows: tuple[int, int], cols: tuple[int, int], d_row_conflicts: int = 0,
                        d_bitmap: int = 0, d_dg: int = 0, d_tm: int = 0):
        """Device-pointer call (raw addresses, e.g. torch.Tensor.data_ptr()); asynchronous."""
        self._check(self.L.msspe_cross_dimer_dev(
            self.ptr, C.c_void_p(d_pool), n, k, C.byref(chem), C.c_float(threshold),
            rows[0], rows[1], cols[0], cols[1], C.c_void_p(d_row_conflicts),
            C.c_void_p(d_bitmap), C.c_void_p(d_dg), C.c_void_p(d_tm)))

    def profile_enable(self, on: bool = True):
        self._check(self.L.msspe_profile_enable(self.ptr, int(on)))

    def profile_read(self) -> tuple[int, float]:
        """(launches of the all-pairs kernel, their summed device time in ms) since the last read."""
        n, ms = C.c_uint64(), C.c_double()
        self._check(self.L.msspe_profile_read(self.ptr, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def last_overflow_pairs(self) -> int:
        v = C.c_uint64()
        self._check(self.L.msspe_last_overflow_pairs(self.ptr, C.byref(v)))
        return int(v.value)

    def segment_coverage(self, seqs: np.ndarray, opt: KmerOpt, fwd: list[str], rev: list[str]) -> np.ndarray:
        """uint8 (n_seq, P): 1 where the segment is covered by the primer set (main.rs:518-594)."""
        a = np.ascontiguousarray(seqs, dtype=np.uint8)
        n_seq, seq_len = a.shape
        P = 0 if seq_len < opt.segment_size else (seq_len - opt.segment_size) // opt.overlap_size + 1
        f = pack_oligos(fwd) if len(fwd) else np.zeros(0, dtype=np.uint64)
        r = pack_oligos(rev) if len(rev) else np.zeros(0, dtype=np.uint64)
        hit = np.zeros((n_seq, P), dtype=np.uint8)
        self._check(self.L.msspe_segment_coverage(
            self.ptr, a.ctypes.data, n_seq, seq_len, C.byref(opt), f.ctypes.data, len(f), r.ctypes.data, len(r),
            hit.ctypes.data))
        return hit

    def pair_stage_samples(self):
        """[(row, col, reason bits)] for up to 1024 pairs the integer stage handed on."""
        v = (C.c_uint64 * 1024)()
        n = C.c_int(0)
        self._check(self.L.msspe_pair_stage_samples(self.ptr, v, 1024, C.byref(n)))
        return [(int(x >> 40), int((x >> 16) & 0xffffff), int(x & 0xffff)) for x in v[:n.value]]

    def pair_stage_stats(self) -> dict:
        """Diagnostics of the exact-integer stages since the last call (resets them): the
        matrix-mode kernel's counts at the top level, the list-mode kernel's under "list";
        "needed_f64" = pairs only the f64 kernels could answer."""
        v = (C.c_uint64 * 16)()
        self._check(self.L.msspe_pair_stage_stats(self.ptr, v))
        names = ("deferred", "tm_near_tie", "loop_eq_value", "loop_tie", "rejected_min", "pick_tie",
                 "replay_mismatch", "path_tie")
        out = {n: int(v[i]) for i, n in enumerate(names)}
        out["list"] = {n: int(v[8 + i]) for i, n in enumerate(names)}
        out["needed_f64"] = int(v[8])
        return out

    # ---- stage B ---------------------------------------------------------------------------
    def oligo_stats(self, pool, chem: Chem | None = None):
        buf, n, k = _ascii(pool)
        chem = chem or Chem.primer3()
        out = {name: np.empty(n) for name in ("tm", "gc", "self_any", "self_end", "hairpin")}
        self._check(self.L.msspe_oligo_stats(self.ptr, buf, n, k, C.byref(chem),
                                             *[out[x].ctypes.data for x in out]))
        return out

    def oligo_stats_dev(self, d_pool: int, n: int, k: int, chem: Chem, d_tm: int = 0, d_gc: int = 0,
                        d_self_any: int = 0, d_self_end: int = 0, d_hairpin: int = 0):
        """Device-pointer call (raw addresses of n packed oligos and of n doubles per requested statistic,
        0 = not wanted); asynchronous on the context's stream."""
        self._check(self.L.msspe_oligo_stats_dev(self.ptr, C.c_void_p(d_pool), n, k, C.byref(chem),
                                                 *[C.c_void_p(x) for x in (d_tm, d_gc, d_self_any, d_self_end, d_hairpin)]))

    # ---- stage A ---------------------------------------------------------------------------
    def kmer_candidates(self, seqs: np.ndarray, opt: KmerOpt, direction: int,
                        device_ptr: int | None = None, n_seq: int | None = None,
                        seq_len: int | None = None, capacity: int | None = None):
        """seqs: uint8 (n_seq, L) host array (or pass device_ptr + shape).  Returns (words, freqs).
        capacity: size of the output buffers (default: max_iterations, which always suffices)."""
        cap = max(1, opt.max_iterations if capacity is None else capacity)
        words = np.zeros(cap, dtype=np.uint64)
        freqs = np.zeros(cap, dtype=np.uint32)
        n_out = C.c_int(0)
        if device_ptr is None:
            a = np.ascontiguousarray(seqs, dtype=np.uint8)
            n_seq, seq_len = a.shape
            self._check(self.L.msspe_kmer_candidates(
                self.ptr, a.ctypes.data, n_seq, seq_len, C.byref(opt), direction,
                words.ctypes.data, freqs.ctypes.data, cap, C.byref(n_out)))
        else:
            self._check(self.L.msspe_kmer_candidates_dev(
                self.ptr, C.c_void_p(device_ptr), n_seq, seq_len, C.byref(opt), direction,
                words.ctypes.data, freqs.ctypes.data, cap, C.byref(n_out)))
        m = n_out.value
        return [unpack_oligo(w, opt.kmer_size) for w in words[:m]], freqs[:m].copy()

    def put_rows_packed(self, seqs: np.ndarray) -> int:
        """Upload an alignment (uint8 (n_seq, L)) in its compact device form (2-bit bases + validity bit, packed on
        the device behind the copy: msspe_device_put_rows_packed).  Returns the device address; free it with
        device_free()."""
        a = np.ascontiguousarray(seqs, dtype=np.uint8)
        n_seq, seq_len = a.shape
        ptrs = (C.c_char_p * n_seq)(*[C.cast(a[i].ctypes.data, C.c_char_p) for i in range(n_seq)])
        lens = (C.c_size_t * n_seq)(*([seq_len] * n_seq))
        dev = C.c_void_p()
        self._check(self.L.msspe_device_put_rows_packed(self.ptr, ptrs, lens, n_seq, seq_len, C.byref(dev)))
        return int(dev.value)

    def device_free(self, device_ptr: int) -> None:
        self._check(self.L.msspe_device_free(self.ptr, C.c_void_p(device_ptr)))

    def kmer_candidates_packed(self, d_packed: int, n_seq: int, seq_len: int, opt: KmerOpt, direction: int,
                               capacity: int | None = None):
        """Stage A on a packed alignment resident on the device (put_rows_packed).  Returns (words, freqs)."""
        cap = max(1, opt.max_iterations if capacity is None else capacity)
        words = np.zeros(cap, dtype=np.uint64)
        freqs = np.zeros(cap, dtype=np.uint32)
        n_out = C.c_int(0)
        self._check(self.L.msspe_kmer_candidates_packed_dev(
            self.ptr, C.c_void_p(d_packed), n_seq, seq_len, C.byref(opt), direction,
            words.ctypes.data, freqs.ctypes.data, cap, C.byref(n_out)))
        m = n_out.value
        return [unpack_oligo(w, opt.kmer_size) for w in words[:m]], freqs[:m].copy()


def _both(self, d_packed: int, n_seq: int, seq_len: int, opt: KmerOpt, capacity: int | None = None):
    """Stage A, both directions of a packed alignment at once (msspe_kmer_candidates_both_packed_dev).
    Returns ((words, freqs) of direction 0, (words, freqs) of direction 1)."""
    cap = max(1, opt.max_iterations if capacity is None else capacity)
    w = [np.zeros(cap, dtype=np.uint64) for _ in range(2)]
    f = [np.zeros(cap, dtype=np.uint32) for _ in range(2)]
    n = [C.c_int(0), C.c_int(0)]
    self._check(self.L.msspe_kmer_candidates_both_packed_dev(
        self.ptr, C.c_void_p(d_packed), n_seq, seq_len, C.byref(opt), w[0].ctypes.data, f[0].ctypes.data, C.byref(n[0]),
        w[1].ctypes.data, f[1].ctypes.data, C.byref(n[1]), cap))
    return tuple(([unpack_oligo(x, opt.kmer_size) for x in w[d][:n[d].value]], f[d][:n[d].value].copy()) for d in (0, 1))


Engine.kmer_candidates_both_packed = _both


def group_rows(n: int, n_members: int, member: int) -> np.ndarray:
    """Pool rows a member of a group screens (include/msspe_hip.h msspe_group_rows; host only)."""
    L = load_library()
    cnt = C.c_int(0)
    rc = L.msspe_group_rows(n, n_members, member, None, 0, C.byref(cnt))
    if rc:
        raise MsspeError(rc, "msspe_group_rows: bad argument")
    rows = np.zeros(max(cnt.value, 1), dtype=np.uint32)
    rc = L.msspe_group_rows(n, n_members, member, rows.ctypes.data, int(rows.size), C.byref(cnt))
    if rc:
        raise MsspeError(rc, "msspe_group_rows failed")
    return rows[:cnt.value]


def rccl_available(library: str | None = None) -> tuple[bool, str]:
    """(loadable with every entry point the group uses, reason if not) -- msspe_group_rccl_available; host only."""
    why = C.create_string_buffer(512)
    ok = load_library().msspe_group_rccl_available(library.encode() if library else None, why, 512)
    return bool(ok), why.value.decode()


class Group:
    """Several devices of one node in one process (include/msspe_hip.h msspe_group_*): one context per listed
    device; a device listed more than once = members sharing a card (the rehearsal mode, transport device-copy)."""

    def __init__(self, devices, params_path: str | None = None, transport: str | None = None):
        self.L = load_library()
        self.ptr = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self.L.msspe_group_create(arr, len(devices), params_path.encode() if params_path else None,
                                       transport.encode() if transport else None, C.byref(self.ptr))
        if rc:
            msg = self.L.msspe_group_last_error(self.ptr).decode() if self.ptr else "allocation failed"
            if self.ptr:
                self.L.msspe_group_destroy(self.ptr)
                self.ptr = C.c_void_p()
            raise MsspeError(rc, msg)

    def close(self):
        if getattr(self, "ptr", None):
            self.L.msspe_group_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise MsspeError(rc, self.L.msspe_group_last_error(self.ptr).decode())

    @property
    def size(self) -> int:
        return int(self.L.msspe_group_size(self.ptr))

    @property
    def transport(self) -> str:
        return self.L.msspe_group_transport(self.ptr).decode()

    @property
    def transport_reason(self) -> str:
        """Why the copies run where transport "auto" wanted RCCL ("" otherwise)."""
        return self.L.msspe_group_transport_reason(self.ptr).decode()

    def set_option(self, key: str, value) -> None:
        self._check(self.L.msspe_group_set_option(self.ptr, key.encode(), str(value).encode()))

    def cross_dimer(self, pool, chem: Chem | None = None, threshold: float = -9000.0, want_bitmap=True):
        buf, n, k = _ascii(pool)
        chem = chem or Chem.ntthal()
        rc_ = np.zeros(n, dtype=np.uint32)
        bm = np.zeros((n, (n + 63) // 64), dtype=np.uint64) if want_bitmap else None
        self._check(self.L.msspe_cross_dimer_group(self.ptr, buf, n, k, C.byref(chem), C.c_float(threshold),
                                                   rc_.ctypes.data, bm.ctypes.data if want_bitmap else None))
        return {"row_conflicts": rc_, "bitmap": bm}

    def cross_dimer_edges(self, pool, chem: Chem | None = None, threshold: float = -9000.0, capacity: int = 1 << 20):
        buf, n, k = _ascii(pool)
        chem = chem or Chem.ntthal()
        edges = np.zeros(capacity, dtype=np.dtype([("a", np.uint32), ("b", np.uint32), ("dg", np.float32)]))
        count = C.c_uint64()
        rc = self.L.msspe_cross_dimer_edges_group(self.ptr, buf, n, k, C.byref(chem), C.c_float(threshold),
                                                  edges.ctypes.data, capacity, C.byref(count))
        if rc:
            err = MsspeError(rc, self.L.msspe_group_last_error(self.ptr).decode())
            err.count = int(count.value)
            raise err
        return edges[:count.value], int(count.value)

    def oligo_stats(self, pool, chem: Chem | None = None):
        buf, n, k = _ascii(pool)
        chem = chem or Chem.primer3()
        out = {name: np.empty(n) for name in ("tm", "gc", "self_any", "self_end", "hairpin")}
        self._check(self.L.msspe_oligo_stats_group(self.ptr, buf, n, k, C.byref(chem),
                                                   *[out[x].ctypes.data for x in out]))
        return out

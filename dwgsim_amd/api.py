"""ctypes binding of the C-ABI in include/dwgsim_hip.h (libdwgsim_hip.so) plus a thin driver that
follows dwgsim_core()'s contig loop (reference src/dwgsim.c:419-1121) over that ABI.

This module is plumbing for tests and bench.py; the product is the shared library.  It never falls
back to a CPU implementation: if the library or a HIP device is missing it raises.
"""
from __future__ import annotations
import ctypes as C
import os
import shlex
from dataclasses import dataclass, field

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DWGSIM_HIP_LIB") or os.path.join(_HERE, "libdwgsim_hip.so")   # env override: analysis builds only

STREAM_BWA1, STREAM_BWA2, STREAM_BFAST = 0, 1, 2
STREAM_NAMES = {0: "bwa.read1.fastq", 1: "bwa.read2.fastq", 2: "bfast.fastq"}


class Params(C.Structure):
    _fields_ = [
        ("e_start", C.c_double * 2), ("e_end", C.c_double * 2),
        ("is_inner", C.c_int32), ("dist", C.c_int32), ("std_dev", C.c_double),
        ("N", C.c_int64), ("C", C.c_double), ("length", C.c_int32 * 2),
        ("mut_rate", C.c_double), ("mut_freq", C.c_double), ("indel_frac", C.c_double), ("indel_extend", C.c_double),
        ("indel_min", C.c_int32), ("rand_read", C.c_double), ("max_n", C.c_int32), ("data_type", C.c_int32),
        ("strandedness", C.c_int32), ("read_one_strand", C.c_int32), ("is_hap", C.c_int32), ("seed", C.c_int32),
        ("fixed_quality", C.c_int32), ("quality_std", C.c_double), ("reads_output_type", C.c_int32),
        ("output_type", C.c_int32), ("amplicons", C.c_int32), ("read_prefix", C.c_char_p), ("flow_order", C.c_char_p),
        ("use_base_error", C.c_int32),
    ]


class Batch(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_random", C.c_uint64), ("n_retries", C.c_uint64), ("bytes", C.c_uint64 * 3),
                ("dev_ptr", C.c_void_p * 3), ("kernel_ms", C.c_float), ("sim_kernel_ms", C.c_float),
                ("fail_seg", C.c_uint64 * 4), ("fail_carry", C.c_uint64), ("gz_bytes", C.c_uint64 * 3)]


class Range(C.Structure):
    """dwgsim_hip_range_t: a read-index range of one contig"""
    _fields_ = [("contig", C.c_int32), ("reserved", C.c_int32), ("first_ii", C.c_uint64), ("n_pairs", C.c_uint64)]


MUT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
READS_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int)
MSG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
READS_AT_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int)      # (ABI 5) pieces with their offset, from several threads


class JobSink(C.Structure):
    """dwgsim_hip_job_sink_t"""
    _fields_ = [("user", C.c_void_p), ("mutations", MUT_CB), ("reads", READS_CB), ("message", MSG_CB), ("reads_at", READS_AT_CB)]


class JobOptions(C.Structure):
    """dwgsim_hip_job_options_t"""
    _fields_ = [("gzip", C.c_int32), ("quiet", C.c_int32), ("batch_pairs", C.c_uint64), ("group_bp", C.c_uint64), ("min_share", C.c_uint64)]


RAND_CHAIN = (1 << 64) - 1        # DWGSIM_HIP_RAND_CHAIN
SKIP_NO_REGION, SKIP_NON_ACGT, SKIP_AMPLICON, SKIP_SHORT_INSERT, SKIP_SHORT_READ, SKIP_NO_PAIRS = -100, -101, -102, -103, -104, -105      # DWGSIM_HIP_SKIP_*


def is_skip(r: int) -> bool:
    """DWGSIM_HIP_IS_SKIP: the value says why dwgsim_core passes over a contig (a range disjoint from the error codes)"""
    return -105 <= r <= -100


EXPORTS = [
    "dwgsim_hip_params_default", "dwgsim_hip_params_check", "dwgsim_hip_pairs_for_contig", "dwgsim_hip_create",
    "dwgsim_hip_destroy", "dwgsim_hip_last_error", "dwgsim_hip_add_contig", "dwgsim_hip_drop_contig",
    "dwgsim_hip_set_regions", "dwgsim_hip_contig_region_length", "dwgsim_hip_contig_set_placement_length",
    "dwgsim_hip_set_mutation_input", "dwgsim_hip_mutate_contig", "dwgsim_hip_mutations_text", "dwgsim_hip_mutations_take", "dwgsim_hip_mutlist_text", "dwgsim_hip_mutlist_free", "dwgsim_hip_count_random", "dwgsim_hip_simulate",
    "dwgsim_hip_fetch", "dwgsim_hip_device_info", "dwgsim_hip_device_numa_node",
    "dwgsim_hip_simulate_async", "dwgsim_hip_wait", "dwgsim_hip_fetch_async", "dwgsim_hip_fetch_wait", "dwgsim_hip_host_alloc", "dwgsim_hip_host_free",
    "dwgsim_hip_add_contigs", "dwgsim_hip_group_layout", "dwgsim_hip_mutate_async", "dwgsim_hip_mutate_wait", "dwgsim_hip_mutate_poll", "dwgsim_hip_count_random_ranges", "dwgsim_hip_simulate_ranges_async", "dwgsim_hip_device_count",
    "dwgsim_hip_job_create", "dwgsim_hip_job_set_contig_table", "dwgsim_hip_job_set_regions", "dwgsim_hip_job_set_mutation_input", "dwgsim_hip_job_prepare", "dwgsim_hip_job_add_contig", "dwgsim_hip_job_begin_contig", "dwgsim_hip_job_commit_contig", "dwgsim_hip_job_cancel_contig", "dwgsim_hip_get_params",
    "dwgsim_hip_job_finish", "dwgsim_hip_job_last_error", "dwgsim_hip_job_destroy",
    "dwgsim_hip_set_fail_carry", "dwgsim_hip_failseg_join", "dwgsim_hip_shard_range", "dwgsim_hip_debug_option", "dwgsim_hip_debug_get", "dwgsim_hip_debug_count_byte", "dwgsim_hip_set_gzip", "dwgsim_hip_fetch_gz_async", "dwgsim_hip_debug_gzip",
    "dwgsim_hip_selftest_fp64", "dwgsim_hip_selftest_lazy", "dwgsim_hip_selftest_text",
]

_lib = None


def load(path: str | None = None):
    """Load the shared library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
    lib = C.CDLL(p)
    P = C.POINTER
    lib.dwgsim_hip_params_default.argtypes = [P(Params)]
    lib.dwgsim_hip_params_check.argtypes = [P(Params), C.c_char_p, C.c_size_t]
    lib.dwgsim_hip_pairs_for_contig.restype = C.c_int64
    lib.dwgsim_hip_pairs_for_contig.argtypes = [P(Params), C.c_int64, C.c_uint64, C.c_int, C.c_int64]
    lib.dwgsim_hip_create.restype = C.c_void_p
    lib.dwgsim_hip_create.argtypes = [P(Params), C.c_int, P(C.c_int)]
    lib.dwgsim_hip_destroy.argtypes = [C.c_void_p]
    lib.dwgsim_hip_last_error.restype = C.c_char_p
    lib.dwgsim_hip_last_error.argtypes = [C.c_void_p]
    lib.dwgsim_hip_add_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_uint32]
    lib.dwgsim_hip_drop_contig.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_add_contigs.argtypes = [C.c_void_p, C.c_int, P(C.c_char_p), P(C.c_void_p), P(C.c_int64), P(C.c_uint32)]
    lib.dwgsim_hip_group_layout.restype = C.c_int64
    lib.dwgsim_hip_group_layout.argtypes = [P(C.c_int64), C.c_int, P(C.c_int64)]
    lib.dwgsim_hip_mutate_async.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_mutate_wait.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_mutate_poll.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_count_random_ranges.argtypes = [C.c_void_p, P(Range), C.c_int, P(C.c_uint64), P(C.c_uint64)]
    lib.dwgsim_hip_simulate_ranges_async.argtypes = [C.c_void_p, P(Range), C.c_int, C.c_uint64, C.c_int]
    lib.dwgsim_hip_device_count.argtypes = []
    lib.dwgsim_hip_job_create.restype = C.c_void_p
    lib.dwgsim_hip_job_create.argtypes = [P(Params), P(C.c_int), C.c_int, P(JobSink), P(JobOptions), P(C.c_int)]
    lib.dwgsim_hip_job_set_contig_table.argtypes = [C.c_void_p, P(C.c_char_p), P(C.c_int64), C.c_int]
    lib.dwgsim_hip_job_set_regions.argtypes = [C.c_void_p, C.c_char_p]
    lib.dwgsim_hip_job_set_mutation_input.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    lib.dwgsim_hip_job_prepare.argtypes = [C.c_void_p, P(C.c_uint64)]
    lib.dwgsim_hip_job_add_contig.restype = C.c_int64
    lib.dwgsim_hip_job_add_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.dwgsim_hip_job_begin_contig.restype = C.c_void_p
    lib.dwgsim_hip_job_begin_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, P(C.c_int64)]
    lib.dwgsim_hip_job_commit_contig.restype = C.c_int64
    lib.dwgsim_hip_job_commit_contig.argtypes = [C.c_void_p]
    lib.dwgsim_hip_job_cancel_contig.argtypes = [C.c_void_p]
    lib.dwgsim_hip_get_params.argtypes = [C.c_void_p, P(Params)]
    lib.dwgsim_hip_job_finish.argtypes = [C.c_void_p]
    lib.dwgsim_hip_job_last_error.restype = C.c_char_p
    lib.dwgsim_hip_job_last_error.argtypes = [C.c_void_p]
    lib.dwgsim_hip_job_destroy.argtypes = [C.c_void_p]
    lib.dwgsim_hip_set_regions.argtypes = [C.c_void_p, C.c_char_p, P(C.c_char_p), P(C.c_int64), C.c_int, P(C.c_uint64)]
    lib.dwgsim_hip_contig_region_length.restype = C.c_int64
    lib.dwgsim_hip_contig_region_length.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int64, P(C.c_int64), P(C.c_int64)]
    lib.dwgsim_hip_contig_set_placement_length.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    lib.dwgsim_hip_set_mutation_input.argtypes = [C.c_void_p, C.c_int, C.c_char_p, P(C.c_char_p), P(C.c_int64), C.c_int]
    lib.dwgsim_hip_mutate_contig.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_mutations_text.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_size_t), P(C.c_void_p), P(C.c_size_t)]
    lib.dwgsim_hip_mutations_take.argtypes = [C.c_void_p, C.c_int, P(C.c_int)]
    lib.dwgsim_hip_mutations_take.restype = C.c_void_p
    lib.dwgsim_hip_mutlist_text.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_size_t), P(C.c_void_p), P(C.c_size_t)]
    lib.dwgsim_hip_mutlist_free.argtypes = [C.c_void_p]
    lib.dwgsim_hip_mutlist_free.restype = None
    lib.dwgsim_hip_count_random.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, P(C.c_uint64)]
    lib.dwgsim_hip_simulate.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, P(Batch)]
    lib.dwgsim_hip_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.dwgsim_hip_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, P(C.c_int), P(C.c_size_t)]
    lib.dwgsim_hip_simulate_async.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    lib.dwgsim_hip_wait.argtypes = [C.c_void_p, C.c_int, P(Batch)]
    lib.dwgsim_hip_fetch_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.dwgsim_hip_fetch_wait.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_host_alloc.restype = C.c_void_p
    lib.dwgsim_hip_host_alloc.argtypes = [C.c_size_t]
    lib.dwgsim_hip_host_free.argtypes = [C.c_void_p]
    lib.dwgsim_hip_set_fail_carry.argtypes = [C.c_void_p, C.c_uint64]
    lib.dwgsim_hip_failseg_join.argtypes = [P(C.c_uint64 * 4), P(C.c_uint64 * 4)]
    lib.dwgsim_hip_shard_range.restype = None
    lib.dwgsim_hip_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, P(C.c_uint64), P(C.c_uint64)]
    lib.dwgsim_hip_debug_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.dwgsim_hip_debug_get.argtypes = [C.c_void_p, C.c_char_p, P(C.c_int64)]
    lib.dwgsim_hip_debug_count_byte.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_uint64)]
    lib.dwgsim_hip_debug_gzip.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, P(C.c_size_t)]
    lib.dwgsim_hip_set_gzip.argtypes = [C.c_void_p, C.c_int]
    lib.dwgsim_hip_fetch_gz_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    if path is None:
        _lib = lib
    return lib


class DwgsimError(RuntimeError):
    pass


def default_params(lib=None) -> Params:
    lib = lib or load()
    p = Params()
    lib.dwgsim_hip_params_default(C.byref(p))
    return p


def _error_rate(s: str):
    """dwgsim_opt.c:162-179 get_error_rate: 'a', 'a-b' or 'a,b'."""
    import re
    m = re.match(r"^([^,\-]*)[,\-]?(.*)$", s)
    start = float(m.group(1))
    sep = None
    for i, ch in enumerate(s):
        if ch in ",-":
            sep = i
            break
    if sep is not None and sep < len(s) - 1:
        return start, float(s[sep + 1:])
    return start, start


def parse_flags(flags: str, lib=None) -> Params:
    """The dwgsim getopt surface (dwgsim_opt.c:211) for the options on the accelerated path."""
    p = default_params(lib)
    toks = shlex.split(flags)
    i = 0
    keep = []
    mut_input = None
    regions = None
    while i < len(toks):
        t = toks[i]
        def arg():
            nonlocal i
            i += 1
            return toks[i]
        if t == "-i": p.is_inner = 1
        elif t == "-d": p.dist = int(arg())
        elif t == "-s": p.std_dev = float(arg())
        elif t == "-N": p.N = int(arg()); p.C = -1
        elif t == "-C": p.C = float(arg()); p.N = -1
        elif t == "-1": p.length[0] = int(arg())
        elif t == "-2": p.length[1] = int(arg())
        elif t == "-e": p.e_start[0], p.e_end[0] = _error_rate(arg())
        elif t == "-E": p.e_start[1], p.e_end[1] = _error_rate(arg())
        elif t == "-r": p.mut_rate = float(arg())
        elif t == "-F": p.mut_freq = float(arg())
        elif t == "-R": p.indel_frac = float(arg())
        elif t == "-X": p.indel_extend = float(arg())
        elif t == "-I": p.indel_min = int(arg())
        elif t == "-c": p.data_type = int(arg())
        elif t == "-S": p.strandedness = int(arg())
        elif t == "-A": p.read_one_strand = int(arg())
        elif t == "-n": p.max_n = int(arg())
        elif t == "-y": p.rand_read = float(arg())
        elif t == "-f": keep.append(arg().encode()); p.flow_order = keep[-1]
        elif t == "-B": p.use_base_error = 1
        elif t == "-H": p.is_hap = 1
        elif t == "-z": p.seed = int(arg())
        elif t == "-M": p.output_type = int(arg())
        elif t == "-P": keep.append(arg().encode()); p.read_prefix = keep[-1]
        elif t == "-q": p.fixed_quality = ord(arg()[0])
        elif t == "-Q": p.quality_std = float(arg())
        elif t == "-o": p.reads_output_type = int(arg())
        elif t == "-a": p.amplicons = 1
        elif t == "-x": regions = arg()
        elif t in ("-m", "-b", "-v"): mut_input = ({"-b": 0, "-m": 1, "-v": 2}[t], arg())
        else:
            raise DwgsimError(f"option {t} is not on the accelerated path")
        i += 1
    p._keep = keep
    p._regions = regions           # -x path, applied by run_job through dwgsim_hip_set_regions
    p._mut_input = mut_input       # (type, path) of -b / -m / -v, applied by run_job through dwgsim_hip_set_mutation_input
    return p


def read_fasta(path: str):
    """seq_read_fasta (mut.c:49-87): name = first token of the header; keeps isalpha, '-' and '.'."""
    import numpy as np
    out = []
    name, chunks = None, []
    with open(path, "rb") as f:
        data = f.read()
    # a '>' opens a record wherever it stands in the sequence, as in the reference's fgetc loop -- but not inside a header line
    at = data.find(b">")
    while at >= 0:
        nl = data.find(b"\n", at)
        header = data[at + 1:] if nl < 0 else data[at + 1:nl]
        nxt = -1 if nl < 0 else data.find(b">", nl + 1)
        body = b"" if nl < 0 else (data[nl + 1:] if nxt < 0 else data[nl + 1:nxt])
        header = header.replace(b"\r", b"")
        name = header.split(b" ")[0].split(b"\t")[0].decode("latin-1")
        arr = np.frombuffer(body, dtype=np.uint8)
        keep = ((arr >= 65) & (arr <= 90)) | ((arr >= 97) & (arr <= 122)) | (arr == 45) | (arr == 46)
        out.append((name, np.ascontiguousarray(arr[keep])))
        at = nxt
    return out


@dataclass
class JobResult:
    streams: dict = field(default_factory=dict)      # stream id -> bytes
    mutations_txt: bytes = b""
    mutations_vcf: bytes = b""
    n_pairs: int = 0
    n_random: int = 0
    n_retries: int = 0
    kernel_ms: float = 0.0
    sim_kernel_ms: float = 0.0
    walk_ms: float = 0.0
    flow_cap_mult: int = 1                           # Ion Torrent: by how much the read capacity had grown at the end of the job (run_job)


VCF_HEADER_POST = (
    b"##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n"
    b"##INFO=<ID=pl,Number=1,Type=Integer,Description=\"Phasing: 1 - HET contig 1, #2 - HET contig #2, 3 - HOM both contigs\">\n"
    b"##INFO=<ID=mt,Number=1,Type=String,Description=\"Variant Type: SUBSTITUTE/INSERT/DELETE\">\n"
    b"#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")


class Context:
    """One GPU context (one per device / per rank)."""

    def __init__(self, params: Params, device: int = 0, lib=None):
        self.lib = lib or load()
        self.params = params
        err = C.c_int(0)
        self.h = self.lib.dwgsim_hip_create(C.byref(params), device, C.byref(err))
        if not self.h:
            raise DwgsimError(f"dwgsim_hip_create failed with code {err.value}" + (" (no HIP device? there is no CPU fallback)" if err.value == -2 else " (the library said why on stderr)"))
        for kv in os.environ.get("DWGSIM_HIP_DEBUG", "").split(","):      # analysis only: "ion_lds=2,flow_slots=3" -> dwgsim_hip_debug_option on every context
            if "=" in kv:
                self.lib.dwgsim_hip_debug_option(self.h, kv.split("=")[0].encode(), int(kv.split("=")[1]))

    def close(self):
        if self.h:
            self.lib.dwgsim_hip_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise DwgsimError(f"error {rc}: {self.lib.dwgsim_hip_last_error(self.h).decode(errors='replace')}")
        return rc

    def add_contig(self, name: str, ascii_arr, contig_index: int) -> int:
        import numpy as np
        arr = np.ascontiguousarray(ascii_arr, dtype=np.uint8)
        return self._chk(self.lib.dwgsim_hip_add_contig(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), len(arr), contig_index))

    def drop_contig(self, cid: int):
        self._chk(self.lib.dwgsim_hip_drop_contig(self.h, cid))

    def add_contigs(self, contigs, first_index: int = 0, indices=None) -> int:
        """contigs: [(name, uint8 array)] resident together as one group; returns the handle of the first (contig k: handle + k)."""
        import numpy as np
        n = len(contigs)
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for _, a in contigs]
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in contigs])
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_int64 * n)(*[len(a) for a in arrs])
        idx = (C.c_uint32 * n)(*(indices if indices is not None else range(first_index, first_index + n)))
        return self._chk(self.lib.dwgsim_hip_add_contigs(self.h, n, names, ptrs, lens, idx))

    def mutate_async(self, cid: int):
        self._chk(self.lib.dwgsim_hip_mutate_async(self.h, cid))

    def mutate_wait(self, cid: int):
        self._chk(self.lib.dwgsim_hip_mutate_wait(self.h, cid))

    @staticmethod
    def _ranges(ranges):
        arr = (Range * len(ranges))()
        for k, (cid, first, n) in enumerate(ranges):
            arr[k].contig, arr[k].first_ii, arr[k].n_pairs = cid, first, n
        return arr

    def count_random_ranges(self, ranges, per_range: bool = False):
        """ranges: [(contig handle, first read index, pairs)] in file order, contigs of one group"""
        n = C.c_uint64(0)
        per = (C.c_uint64 * len(ranges))() if per_range else None
        self._chk(self.lib.dwgsim_hip_count_random_ranges(self.h, self._ranges(ranges), len(ranges), C.byref(n), per))
        return list(per) if per_range else n.value

    def simulate_ranges_async(self, ranges, rand_base: int = RAND_CHAIN, slot: int = 0):
        self._chk(self.lib.dwgsim_hip_simulate_ranges_async(self.h, self._ranges(ranges), len(ranges), rand_base, slot))

    def simulate_ranges(self, ranges, rand_base: int, slot: int = 0) -> Batch:
        self.simulate_ranges_async(ranges, rand_base, slot)
        return self.wait(slot)

    def set_regions(self, path: str, contigs) -> int:
        n = len(contigs)
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in contigs])
        lens = (C.c_int64 * n)(*[int(a) if isinstance(a, int) else len(a) for _, a in contigs])
        tot = C.c_uint64(0)
        self._chk(self.lib.dwgsim_hip_set_regions(self.h, path.encode(), names, lens, n, C.byref(tot)))
        return tot.value

    def region_length(self, contig_index: int, ascii_arr) -> int:
        import numpy as np
        arr = np.ascontiguousarray(ascii_arr, dtype=np.uint8)
        return self.lib.dwgsim_hip_contig_region_length(self.h, contig_index, arr.ctypes.data_as(C.c_void_p), len(arr), None, None)

    def set_placement_length(self, cid: int, l: int):
        self._chk(self.lib.dwgsim_hip_contig_set_placement_length(self.h, cid, l))

    def set_mutation_input(self, mtype: int, path: str, contigs):
        """contigs: the FASTA's (name, array-or-length) list in file order."""
        n = len(contigs)
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in contigs])
        lens = (C.c_int64 * n)(*[int(a) if isinstance(a, int) else len(a) for _, a in contigs])
        self._chk(self.lib.dwgsim_hip_set_mutation_input(self.h, mtype, path.encode(), names, lens, n))

    def mutate(self, cid: int):
        self._chk(self.lib.dwgsim_hip_mutate_contig(self.h, cid))

    def mutations_text(self, cid: int):
        t, v = C.c_void_p(), C.c_void_p()
        tl, vl = C.c_size_t(), C.c_size_t()
        self._chk(self.lib.dwgsim_hip_mutations_text(self.h, cid, C.byref(t), C.byref(tl), C.byref(v), C.byref(vl)))
        return C.string_at(t, tl.value) if tl.value else b"", C.string_at(v, vl.value) if vl.value else b""

    def mutations_via_list(self, cid: int):
        """the same text through the two-halves form (mutations_take / mutlist_text): [(txt, vcf)] for every contig of cid's group"""
        n = C.c_int()
        L = self.lib.dwgsim_hip_mutations_take(self.h, cid, C.byref(n))
        if not L:
            raise RuntimeError(self.lib.dwgsim_hip_last_error(self.h).decode(errors="replace"))
        try:
            out = []
            for k in range(n.value):
                t, v = C.c_void_p(), C.c_void_p()
                tl, vl = C.c_size_t(), C.c_size_t()
                self._chk(self.lib.dwgsim_hip_mutlist_text(L, k, C.byref(t), C.byref(tl), C.byref(v), C.byref(vl)))
                out.append((C.string_at(t, tl.value) if tl.value else b"", C.string_at(v, vl.value) if vl.value else b""))
            return out
        finally:
            self.lib.dwgsim_hip_mutlist_free(L)

    def count_random(self, cid: int, first_ii: int, n_pairs: int) -> int:
        n = C.c_uint64(0)
        self._chk(self.lib.dwgsim_hip_count_random(self.h, cid, first_ii, n_pairs, C.byref(n)))
        return n.value

    def simulate(self, cid: int, first_ii: int, n_pairs: int, rand_base: int, slot: int = 0) -> Batch:
        b = Batch()
        self._chk(self.lib.dwgsim_hip_simulate(self.h, cid, first_ii, n_pairs, rand_base, slot, C.byref(b)))
        return b

    def simulate_async(self, cid: int, first_ii: int, n_pairs: int, rand_base: int = RAND_CHAIN, slot: int = 0):
        self._chk(self.lib.dwgsim_hip_simulate_async(self.h, cid, first_ii, n_pairs, rand_base, slot))

    def wait(self, slot: int = 0) -> Batch:
        b = Batch()
        self._chk(self.lib.dwgsim_hip_wait(self.h, slot, C.byref(b)))
        return b

    def fetch_async(self, slot: int, stream: int, host_ptr: int, cap: int):
        self._chk(self.lib.dwgsim_hip_fetch_async(self.h, slot, stream, host_ptr, cap))

    def fetch_wait(self, slot: int):
        self._chk(self.lib.dwgsim_hip_fetch_wait(self.h, slot))

    def set_fail_carry(self, carry: int):
        self._chk(self.lib.dwgsim_hip_set_fail_carry(self.h, carry))

    def debug_gzip(self, data: bytes) -> bytes:
        """Test hook: the gzip kernel's members for arbitrary bytes."""
        cap = len(data) + len(data) // 8 + 512 * (len(data) // 32768 + 1) + 64
        out = C.create_string_buffer(cap)
        n = C.c_size_t(0)
        self._chk(self.lib.dwgsim_hip_debug_gzip(self.h, data, len(data), out, cap, C.byref(n)))
        return out.raw[:n.value]

    def set_gzip(self, on: bool = True):
        self._chk(self.lib.dwgsim_hip_set_gzip(self.h, 1 if on else 0))

    def fetch_gz(self, slot: int, stream: int, nbytes: int) -> bytes:
        """The .gz form of a finished stream (GPU gzip), through a page-locked bounce buffer."""
        if not nbytes:
            return b""
        p = self.lib.dwgsim_hip_host_alloc(int(nbytes))
        try:
            self._chk(self.lib.dwgsim_hip_fetch_gz_async(self.h, slot, stream, p, int(nbytes)))
            self._chk(self.lib.dwgsim_hip_fetch_wait(self.h, slot))
            return C.string_at(p, int(nbytes))
        finally:
            self.lib.dwgsim_hip_host_free(p)

    def count_byte(self, slot: int, stream: int, byte: int) -> int:
        n = C.c_uint64(0)
        self._chk(self.lib.dwgsim_hip_debug_count_byte(self.h, slot, stream, byte, C.byref(n)))
        return n.value

    def fetch_np(self, slot: int, stream: int, nbytes: int):
        """The stream as a numpy uint8 array (page-locked staging inside the library)."""
        import numpy as np
        out = np.empty(int(nbytes), dtype=np.uint8)
        self._chk(self.lib.dwgsim_hip_fetch(self.h, slot, stream, out.ctypes.data_as(C.c_void_p), int(nbytes)))
        return out

    def debug_get(self, key: str) -> int:
        v = C.c_int64(0)
        self._chk(self.lib.dwgsim_hip_debug_get(self.h, key.encode(), C.byref(v)))
        return v.value

    def debug_option(self, key: str, value: int):
        self._chk(self.lib.dwgsim_hip_debug_option(self.h, key.encode(), value))

    def fetch(self, slot: int, stream: int, nbytes: int) -> bytes:
        buf = C.create_string_buffer(int(nbytes) if nbytes else 1)
        self._chk(self.lib.dwgsim_hip_fetch(self.h, slot, stream, buf, nbytes))
        return buf.raw[:nbytes]


def pairs_for_contig(params: Params, l: int, tot_len: int, is_last: bool, n_sim: int, lib=None) -> int:
    lib = lib or load()
    return lib.dwgsim_hip_pairs_for_contig(C.byref(params), l, tot_len, 1 if is_last else 0, n_sim)


def shard_range(n_pairs: int, rank: int, world: int, lib=None):
    lib = lib or load()
    first, n = C.c_uint64(0), C.c_uint64(0)
    lib.dwgsim_hip_shard_range(n_pairs, rank, world, C.byref(first), C.byref(n))
    return first.value, n.value


def schedule_contigs(params: Params, contigs, ctx, lib=None):
    """The scheduling half of dwgsim_core's contig loop (dwgsim.c:519-625): which contigs are simulated, with how many pairs and which
    placement length.  Yields (contig ordinal, name, array, n_pairs, l_eff)."""
    lib = lib or load()
    tot_len = sum(len(a) for _, a in contigs)
    want_reads = params.output_type != 2
    have_regions = bool(getattr(params, "_regions", None))
    if have_regions:
        tot_len = ctx.set_regions(params._regions, contigs)          # dwgsim.c:499-506
    n_sim = 0
    n_ref = len(contigs)
    for ci, (name, arr) in enumerate(contigs):
        n_ref -= 1
        n_pairs = 0
        l_eff = len(arr)
        if want_reads:
            last_takes_rest = n_ref == 0 and params.C < 0               # dwgsim.c:535-537: no region bookkeeping on this path
            if have_regions and not last_takes_rest:
                l_eff = ctx.region_length(ci, arr)                      # dwgsim.c:539-581
                if l_eff < 0:
                    continue                                            # skip #0 / #1
            n_pairs = pairs_for_contig(params, l_eff, tot_len, n_ref == 0, n_sim, lib)
            if n_pairs < 0:
                continue                      # skip rules #2-#5: no mutations either (dwgsim.c:596-623)
        n_sim += n_pairs
        yield ci, name, arr, n_pairs, l_eff


def split_ranges(ranges, batch_pairs):
    """[(contig, first, n)] in file order -> batches of at most batch_pairs pairs (a batch may hold several ranges and end inside a contig)"""
    batch, room = [], batch_pairs
    for cid, first, n in ranges:
        while n > 0:
            take = min(n, room)
            batch.append((cid, first, take))
            first += take; n -= take; room -= take
            if room == 0:
                yield batch
                batch, room = [], batch_pairs
    if batch:
        yield batch


def run_job(params: Params, contigs, device: int = 0, batch_pairs: int = 1 << 22, fetch: bool = True, lib=None, debug_options=None, group_bp: int = 0, check_list_form: bool = False) -> JobResult:
    """dwgsim_core (dwgsim.c:419-1121) over the C-ABI: header pass, then schedule -> mutate -> mutations text -> simulate in
    read-index batches.  group_bp = 0: contig after contig, as the reference walks them.  group_bp > 0: consecutive contigs are resident
    together in groups of up to group_bp bases (dwgsim_hip_add_contigs): one walk per group, batches that run across contig boundaries."""
    lib = lib or load()
    res = JobResult(streams={0: bytearray(), 1: bytearray(), 2: bytearray()})
    want_mut = params.output_type != 1
    want_reads = params.output_type != 2
    vcf = bytearray()
    txt = bytearray()
    if want_mut:
        vcf += b"##fileformat=VCFv4.1\n"
        for name, arr in contigs:
            vcf += f"##contig=<ID={name},length={len(arr)}>\n".encode()
        vcf += VCF_HEADER_POST
    n_sim = 0
    rand_ii = 0
    with Context(params, device, lib) as ctx:
        for k, v in (debug_options or {}).items():
            ctx.debug_option(k, v)
        if getattr(params, "_mut_input", None):
            ctx.set_mutation_input(params._mut_input[0], params._mut_input[1], contigs)
        have_regions = bool(getattr(params, "_regions", None))
        sched = list(schedule_contigs(params, contigs, ctx, lib))
        groups, cur, cur_bp = [], [], 0
        for ent in sched:
            if cur and (group_bp <= 0 or cur_bp + len(ent[2]) > group_bp):
                groups.append(cur); cur, cur_bp = [], 0
            cur.append(ent); cur_bp += len(ent[2])
        if cur:
            groups.append(cur)
        for grp in groups:
            h0 = ctx.add_contigs([(name, arr) for _, name, arr, _, _ in grp], indices=[ci for ci, _, _, _, _ in grp])
            if have_regions:
                for k, ent in enumerate(grp):
                    ctx.set_placement_length(h0 + k, ent[4])
            ctx.mutate(h0)
            if want_mut:
                for k in range(len(grp)):
                    t, v = ctx.mutations_text(h0 + k)
                    txt += t
                    vcf += v
                if check_list_form:      # (tests) the two-halves form the job level uses must give the same text
                    tv = ctx.mutations_via_list(h0)
                    if b"".join(a for a, _ in tv) != b"".join(ctx.mutations_text(h0 + k)[0] for k in range(len(grp))) or \
                       b"".join(b for _, b in tv) != b"".join(ctx.mutations_text(h0 + k)[1] for k in range(len(grp))):
                        raise DwgsimError("mutations_take / mutlist_text differs from mutations_text")
            ranges = [(h0 + k, 0, ent[3]) for k, ent in enumerate(grp) if want_reads and ent[3] > 0]
            for batch in split_ranges(ranges, batch_pairs):
                b = ctx.simulate_ranges(batch, rand_ii, 0)
                res.kernel_ms += b.kernel_ms
                res.sim_kernel_ms += b.sim_kernel_ms
                res.n_retries += b.n_retries
                if fetch:
                    for s in range(3):
                        if b.bytes[s]:
                            res.streams[s] += ctx.fetch(0, s, b.bytes[s])
                rand_ii += b.n_random
                n_sim += b.n_pairs
            ctx.drop_contig(h0)
        res.flow_cap_mult = ctx.debug_get("flow_cap_mult")
    res.n_pairs = n_sim
    res.n_random = rand_ii
    res.mutations_txt = bytes(txt)
    res.mutations_vcf = bytes(vcf)
    res.streams = {k: bytes(v) for k, v in res.streams.items()}
    return res


def run_job_api(params: Params, contigs, devices=None, gzip_on_gpu: bool = True, batch_pairs: int = 0, group_bp: int = 0, min_share: int = 0,
                lib=None, keep_output: bool = True, offset_sink: bool = False) -> JobResult:
    """The same job through the JOB level of the C-ABI (dwgsim_hip_job_*): the library schedules, groups, shards over `devices`
    (default: all) and delivers in file order; the sink below only collects.  Streams are returned as text (gzip members are
    decompressed here)."""
    import gzip as _gz
    import numpy as np
    lib = lib or load()
    res = JobResult(streams={0: bytearray(), 1: bytearray(), 2: bytearray()})
    raw = {0: [], 1: [], 2: []}
    txt, vcf = bytearray(), bytearray()
    if params.output_type != 1:
        vcf += b"##fileformat=VCFv4.1\n"
        for name, arr in contigs:
            vcf += f"##contig=<ID={name},length={len(arr)}>\n".encode()
        vcf += VCF_HEADER_POST
    order = {"mut": [], "text_n": {0: 0, 1: 0, 2: 0}}

    def on_mut(user, name, t, tl, v, vl):
        order["mut"].append(name.decode())
        txt.extend(C.string_at(t, tl) if tl else b"")
        vcf.extend(C.string_at(v, vl) if vl else b"")
        return 0

    def on_reads(user, stream, data, n, text_n, gz):
        order["text_n"][stream] += text_n
        if keep_output:
            raw[stream].append((bool(gz), C.string_at(data, n), text_n))
        return 0

    import threading
    at_lock = threading.Lock(); at_pieces = {0: [], 1: [], 2: []}; at_threads = set()

    def on_reads_at(user, stream, offset, data, n, text_n, gz):      # (called from several threads of the job: one per device and stream)
        blob = C.string_at(data, n) if keep_output else b""
        with at_lock:
            at_threads.add(threading.get_ident())
            order["text_n"][stream] += text_n
            at_pieces[stream].append((offset, n, bool(gz), blob, text_n))
        return 0

    sink = JobSink(None, MUT_CB(on_mut), READS_CB(on_reads), MSG_CB(lambda u, m: None), READS_AT_CB(on_reads_at) if offset_sink else READS_AT_CB())
    opt = JobOptions(1 if gzip_on_gpu else 0, 1, batch_pairs, group_bp, min_share)
    err = C.c_int(0)
    devs = (C.c_int * len(devices))(*devices) if devices else None
    job = lib.dwgsim_hip_job_create(C.byref(params), devs, len(devices) if devices else 0, C.byref(sink), C.byref(opt), C.byref(err))
    if not job:
        raise DwgsimError(f"dwgsim_hip_job_create failed with code {err.value}")
    try:
        def chk(rc):
            if rc < 0:
                raise DwgsimError(f"error {rc}: {lib.dwgsim_hip_job_last_error(job).decode(errors='replace')}")
        n = len(contigs)
        names = (C.c_char_p * n)(*[nm.encode() for nm, _ in contigs])
        lens = (C.c_int64 * n)(*[len(a) for _, a in contigs])
        chk(lib.dwgsim_hip_job_set_contig_table(job, names, lens, n))
        if getattr(params, "_regions", None):
            chk(lib.dwgsim_hip_job_set_regions(job, params._regions.encode()))
        if getattr(params, "_mut_input", None):
            chk(lib.dwgsim_hip_job_set_mutation_input(job, params._mut_input[0], params._mut_input[1].encode()))
        chk(lib.dwgsim_hip_job_prepare(job, None))
        for name, arr in contigs:
            a = np.ascontiguousarray(arr, dtype=np.uint8)
            r = lib.dwgsim_hip_job_add_contig(job, name.encode(), a.ctypes.data_as(C.c_void_p), len(a))
            if r < 0 and not is_skip(r):      # a real error: stop feeding the job
                chk(r)
            if r > 0:
                res.n_pairs += r
        chk(lib.dwgsim_hip_job_finish(job))
    finally:
        lib.dwgsim_hip_job_destroy(job)
    if offset_sink:      # the pieces must tile [0, total) of every stream, each exactly once; put in order they are the stream
        for s_ in range(3):
            at = 0
            for off, n, gz, blob, text_n in sorted(at_pieces[s_]):
                if off != at:
                    raise DwgsimError(f"reads_at: stream {s_} has a gap or an overlap at offset {at} (next piece at {off})")
                at += n
                raw[s_].append((gz, blob, text_n))
        res.delivery_threads = len(at_threads)
    for s_ in range(3):
        for gz, blob, text_n in raw[s_]:
            piece = _gz.decompress(blob) if gz else blob
            assert len(piece) == text_n
            res.streams[s_] += piece
    res.mutations_txt, res.mutations_vcf = bytes(txt), bytes(vcf)
    res.streams = {k: bytes(v) for k, v in res.streams.items()}
    res.delivered_text_bytes = dict(order["text_n"])
    return res

"""ctypes binding of include/bbp.h. Fails loudly when the HIP library is missing: there is no fallback."""
import ctypes
import os

# four engine streams per context = HIP's default number of hardware queues; one more active stream in the process and two of
# them share a queue.  Only effective if the HIP runtime has not initialised yet; harmless otherwise.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

_HERE = os.path.dirname(os.path.abspath(__file__))
# BBP_LIB_VARIANT=name loads libbbp_hip.name.so: experiment builds of the same sources with other -D knobs (tools/build_variant.py)
_VARIANT = os.environ.get("BBP_LIB_VARIANT", "")
lib_path = os.path.join(_HERE, "libbbp_hip.%s.so" % _VARIANT if _VARIANT else "libbbp_hip.so")

STATUS = {0: "OK", 1: "VERIFY", 2: "GENS_LEN", 3: "FORMAT", 4: "BAD_ARG", 5: "DEVICE", 6: "INTERNAL"}
STREAM_CONTEXT = ctypes.c_void_p(-1).value  # BBP_STREAM_CONTEXT: the context's own stream; 0 / None-less handles are the caller's hipStream_t
LAYOUT_BLIND_G_H, LAYOUT_BLIND_G = 0, 1
BASE_BBLIND, BASE_G0, BASE_H0, BASE_B, NUM_BASES = 0, 1, 2049, 4097, 4098
R1CS_PROOF_BYTES = 1121

# every symbol include/bbp.h declares: (restype, argtypes)
_vp, _u32, _i32, _u64, _cp = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_char_p
SIGNATURES = {
    "bbp_init": (_i32, [_i32, ctypes.POINTER(_vp)]),
    "bbp_free": (None, [_vp]),
    "bbp_init_all": (_i32, [ctypes.POINTER(_vp)]),
    "bbp_pool_init": (_i32, [ctypes.POINTER(_i32), _u32, ctypes.POINTER(_vp)]),
    "bbp_pool_size": (_u32, [_vp]),
    "bbp_pool_member": (_vp, [_vp, _u32]),
    "bbp_pool_member_stats": (_i32, [_vp, _u32, ctypes.POINTER(_u64), ctypes.POINTER(_u64)]),
    "bbp_last_error": (_cp, [_vp]),
    "bbp_context_stream": (_vp, [_vp]),
    "bbp_context_copy_stream": (_vp, [_vp]),
    "bbp_context_verify_stream": (_vp, [_vp, _u32]),
    "bbp_get_generator": (_i32, [_vp, _u32, _vp]),
    "bbp_get_mimc_constant": (_i32, [_vp, _u32, _vp]),
    "bbp_msm_batch": (_i32, [_vp, _u32, _u32, _vp, _u32, _vp]),
    "bbp_msm_batch_dev": (_i32, [_vp, _u32, _u32, _vp, _u32, _vp, _vp]),
    "bbp_witness_batch": (_i32, [_vp, _u32, _vp, _vp]),
    "bbp_prove": (_i32, [_vp, _vp, _vp, _u32, _u64, _vp, _vp, ctypes.POINTER(_u32)]),
    "bbp_prove_async": (_i32, [_vp, _vp, _vp, _u32, _u64, _vp, _vp, _vp, _vp]),
    "bbp_verify_async": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    "bbp_proof_record_size": (_u32, [_u32]),
    "bbp_entropy_size": (_u32, [_u32]),
    "bbp_verify": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32]),
    "bbp_prove_batch": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "bbp_verify_batch": (_i32, [_vp, _u32, _u32, _vp, _vp]),
    "bbp_prove_batch_dev": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "bbp_verify_batch_dev": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "bbp_prepare_bids_dev": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bbp_verify_batch_aggregated": (_i32, [_vp, _u32, _u32, _vp, _vp, _u32, _vp]),
    "bbp_verify_batch_aggregated_dev": (_i32, [_vp, _u32, _u32, _vp, _vp, _vp, _u32, _vp, _vp]),
    "bbp_reserve": (_i32, [_vp, _u32, _u32]),
    "bbp_set_batching": (_i32, [_vp, _u32, _u32]),
    "bbp_batching_stats": (_i32, [_vp, ctypes.POINTER(_u64), ctypes.POINTER(_u64), ctypes.POINTER(_u32)]),
    "bbp_debug_compile_circuit": (_i32, [_u32, ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "bbp_check_health": (_i32, [_vp, ctypes.POINTER(_u32)]),
    "bbp_debug_corrupt_scratch": (_i32, [_vp]),
    "bbp_describe": (_i32, [_vp, _vp, _u32]),
    "bbp_debug_challenges": (_i32, [_vp, _u32, _u32, _u32, _vp]),
    "bbp_ubench": (_i32, [_vp, _i32, _u32, _u32, ctypes.POINTER(ctypes.c_double)]),
    "bbp_set_profiling": (_i32, [_vp, _i32]),
    "bbp_last_timings": (_i32, [_vp, _vp, _u32, ctypes.POINTER(_u32)]),
}


DONE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int32)  # bbp_done_fn


class BbpError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("bbp status %d (%s) %s" % (status, STATUS.get(status, "?"), msg))
        self.status = status


def _load():
    if not os.path.exists(lib_path):
        raise ImportError(os.path.basename(lib_path) + " not built (run `python __graft_entry__.py`); there is no CPU fallback")
    L = ctypes.CDLL(lib_path)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)  # AttributeError here = the .so does not export what include/bbp.h declares
        f.restype, f.argtypes = res, args
    return L


lib = _load()


def record_size(n):
    return R1CS_PROOF_BYTES + 32 * (4 + n)


def entropy_size(n):
    return 32 * (4 + n) + 32


def _buf(b):
    return (ctypes.c_uint8 * len(b)).from_buffer_copy(bytes(b))


def _stream(s):
    """None -> the context's own stream (BBP_STREAM_CONTEXT); an int is a hipStream_t handle (0 = the legacy default stream)."""
    return STREAM_CONTEXT if s is None else s


def compile_circuit(n_items):
    """Host-only synthesis of the blind-bid circuit for a list length (no device): (status, n_mul, n_cons)."""
    a, b = ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.bbp_debug_compile_circuit(n_items, ctypes.byref(a), ctypes.byref(b))
    return rc, a.value, b.value


class Context:
    """One engine context per GPU (bbp_init). Mirrors the reference's implicit process-wide state
    (lazy_static CONSTANTS + generate_cs_transcript, src/blindbid/mod.rs:7-40) as an explicit handle."""

    def __init__(self, device=0, _borrowed=None):
        if _borrowed is not None:  # a pool member: owned by its pool, never freed from here
            self._h, self._owned = ctypes.c_void_p(_borrowed), False
            return
        self._owned = True
        self._h = ctypes.c_void_p()
        rc = self._init(device)
        if rc != 0:
            msg = lib.bbp_last_error(self._h).decode() if self._h else "no usable HIP device"
            if self._h:
                lib.bbp_free(self._h)
                self._h = None
            raise BbpError(rc, msg)

    def _init(self, device):
        return lib.bbp_init(device, ctypes.byref(self._h))

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                lib.bbp_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BbpError(rc, lib.bbp_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        """hipStream_t handle of the context's own stream (wrap it with torch.cuda.ExternalStream to enqueue torch work on it)."""
        return lib.bbp_context_stream(self._h)

    @property
    def copy_stream(self):
        """hipStream_t handle of the context's idle second stream, for the caller's ingest copies / bbp_prepare_bids_dev."""
        return lib.bbp_context_copy_stream(self._h)

    def generator(self, index):
        out = (ctypes.c_uint8 * 32)()
        self._check(lib.bbp_get_generator(self._h, index, out))
        return bytes(out)

    def mimc_constant(self, i):
        out = (ctypes.c_uint8 * 32)()
        self._check(lib.bbp_get_mimc_constant(self._h, i, out))
        return bytes(out)

    def msm_batch(self, B, n_terms, scalars, layout):
        assert len(scalars) == B * n_terms * 32
        out = (ctypes.c_uint8 * (32 * B))()
        self._check(lib.bbp_msm_batch(self._h, B, n_terms, _buf(scalars), layout, out))
        return bytes(out)

    def msm_batch_dev(self, B, n_terms, scalars_ptr, layout, out_ptr, stream=None):
        self._check(lib.bbp_msm_batch_dev(self._h, B, n_terms, scalars_ptr, layout, out_ptr, _stream(stream)))

    def witness_batch(self, dks):
        B = len(dks) // 96
        out = (ctypes.c_uint8 * (192 * B))()
        self._check(lib.bbp_witness_batch(self._h, B, _buf(dks), out))
        return bytes(out)

    def prove(self, scalars7, pub_list, toggle, entropy=None):
        """Proof::prove (src/blindbid/proof.rs:36-46) -> raw record (R1CSProof || commitments || t_c)."""
        n = len(pub_list) // 32
        out = (ctypes.c_uint8 * record_size(max(n, 1)))()
        plen = ctypes.c_uint32()
        ent = _buf(entropy) if entropy is not None else None
        self._check(lib.bbp_prove(self._h, _buf(scalars7), _buf(pub_list) if n else None, n, toggle, ent, out,
                                  ctypes.byref(plen)))
        return bytes(out)[:record_size(n)]

    def verify(self, record, score, z_img, seed, pub_list):
        """Verify::verify (src/blindbid/verify.rs:47): returns the bbp_status (0 = Ok(()))."""
        n = len(pub_list) // 32
        return lib.bbp_verify(self._h, _buf(record), len(record), _buf(score), _buf(z_img), _buf(seed),
                              _buf(pub_list) if n else None, n)

    def prove_async(self, scalars7, pub_list, toggle, entropy, on_done):
        """bbp_prove_async: returns at once; on_done(status, record_bytes_or_None) is called on an engine thread.  The returned
        object keeps the buffers and the ctypes callback alive: hold it until on_done has run."""
        n = len(pub_list) // 32
        out = (ctypes.c_uint8 * record_size(max(n, 1)))()

        def _cb(_user, status):
            on_done(status, bytes(out)[:record_size(n)] if status == 0 else None)
        cb = DONE_FN(_cb)
        ent = _buf(entropy) if entropy is not None else None
        self._check(lib.bbp_prove_async(self._h, _buf(scalars7), _buf(pub_list), n, toggle, ent, out, cb, None))
        return (cb, out)

    def verify_async(self, record, score, z_img, seed, pub_list, on_done):
        """bbp_verify_async: on_done(status) on an engine thread -- or at once, from this call, when the host-side structural
        parse already decides (the C call then returns the status and never calls back)."""
        n = len(pub_list) // 32
        cb = DONE_FN(lambda _user, status: on_done(status))
        rc = lib.bbp_verify_async(self._h, _buf(record), len(record), _buf(score), _buf(z_img), _buf(seed), _buf(pub_list), n, cb, None)
        if rc != 0:
            on_done(rc)
        return cb

    def prove_batch(self, B, N, inputs, entropy=None):
        out = (ctypes.c_uint8 * (B * record_size(N)))()
        status = (ctypes.c_int32 * B)()
        ent = _buf(entropy) if entropy is not None else None
        self._check(lib.bbp_prove_batch(self._h, B, N, _buf(inputs), ent, out, status))
        return bytes(out), list(status)

    def verify_batch(self, B, N, inputs):
        status = (ctypes.c_int32 * B)()
        self._check(lib.bbp_verify_batch(self._h, B, N, _buf(inputs), status))
        return list(status)

    def prepare_bids_dev(self, B, N, bids_ptr, lists_ptr, toggles_ptr, prove_in_ptr, verify_tail_ptr=None, stream=None):
        """Device-side caller pass: (d,k,seed) + bid list + toggle -> rows for prove_batch_dev / the tail of verify_batch_dev rows."""
        self._check(lib.bbp_prepare_bids_dev(self._h, B, N, bids_ptr, lists_ptr, toggles_ptr, prove_in_ptr, verify_tail_ptr, _stream(stream)))

    def verify_batch_aggregated(self, B, N, inputs, group=0):
        """Statuses as verify_batch; proofs are checked in groups of `group` (0 = default 32) with one generator MSM per group,
        failing groups proof by proof.  Returns (statuses, number of proofs that took the per-proof path)."""
        status = (ctypes.c_int32 * B)()
        nfb = ctypes.c_uint32()
        self._check(lib.bbp_verify_batch_aggregated(self._h, B, N, _buf(inputs), status, group, ctypes.byref(nfb)))
        return list(status), nfb.value

    def verify_batch_aggregated_dev(self, B, N, in_ptr, ent_ptr, status_ptr, group=0, stream=None, want_count=True):
        """Stream-ordered unless the fallback count is asked for (delivering it synchronises the stream): want_count=False
        returns None and leaves the call asynchronous like verify_batch_dev."""
        nfb = ctypes.c_uint32()
        self._check(lib.bbp_verify_batch_aggregated_dev(self._h, B, N, in_ptr, ent_ptr, status_ptr, group,
                                                        ctypes.byref(nfb) if want_count else None, _stream(stream)))
        return nfb.value if want_count else None

    def prove_batch_dev(self, B, N, in_ptr, ent_ptr, out_ptr, stream=None):
        self._check(lib.bbp_prove_batch_dev(self._h, B, N, in_ptr, ent_ptr, out_ptr, _stream(stream)))

    def verify_batch_dev(self, B, N, in_ptr, ent_ptr, status_ptr, stream=None):
        self._check(lib.bbp_verify_batch_dev(self._h, B, N, in_ptr, ent_ptr, status_ptr, _stream(stream)))

    def verify_stream(self, lane):
        """hipStream_t handle of verifier lane 0 / 1: verification calls on the two lanes' streams overlap on the device."""
        return lib.bbp_context_verify_stream(self._h, lane)

    def reserve(self, max_batch, N):
        """Grow every per-batch buffer to what batches of up to max_batch items of list length N need (bbp_reserve)."""
        self._check(lib.bbp_reserve(self._h, max_batch, N))

    def set_batching(self, window_us=0, max_batch=0):
        """Micro-batching window / size bound of the call combiner (concurrent prove() / verify() callers share device batches)."""
        self._check(lib.bbp_set_batching(self._h, window_us, max_batch))

    def batching_stats(self):
        """(combined device calls, requests they carried, largest batch) since the context was created."""
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint32()
        self._check(lib.bbp_batching_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def health(self):
        """0 = healthy; bit 0 = an MSM gather had to be clamped since the context was created (include/bbp.h bbp_check_health)."""
        f = ctypes.c_uint32()
        self._check(lib.bbp_check_health(self._h, ctypes.byref(f)))
        return f.value

    def describe(self):
        """Text report of the device and configuration; lines starting with WARNING: name conditions that cost throughput."""
        buf = ctypes.create_string_buffer(8192)
        self._check(lib.bbp_describe(self._h, buf, len(buf)))
        return buf.value.decode()

    def debug_corrupt_scratch(self):
        """Test hook: the next MSM launch finds an out-of-range entry in its sorted scratch (bbp_debug_corrupt_scratch)."""
        self._check(lib.bbp_debug_corrupt_scratch(self._h))

    def debug_challenges(self, B, N, proof):
        out = (ctypes.c_uint8 * (32 * 32))()
        self._check(lib.bbp_debug_challenges(self._h, B, N, proof, out))
        raw = bytes(out)
        names = ["y", "z", "u", "x", "w", "y_inv", "t1", "t2", "t3", "t4", "t5", "t6", "tb1", "tb2", "tb3", "tb4", "tb5", "tb6",
                 "t_x", "t_x_blinding", "e_blinding", "uj", "uji", "a", "b", "r", "allinv", "wc", "delta"]
        return {n: raw[32 * i:32 * i + 32].hex() for i, n in enumerate(names)}

    def ubench(self, kind, blocks=4096, iters=2000):
        v = ctypes.c_double()
        self._check(lib.bbp_ubench(self._h, kind, blocks, iters, ctypes.byref(v)))
        return v.value

    def set_profiling(self, on):
        self._check(lib.bbp_set_profiling(self._h, 1 if on else 0))

    def last_timings(self, cap=1 << 16):
        """[(tag, microseconds)] for every kernel launched since the last drain (profiling must be on)."""
        arr = (ctypes.c_float * cap)()
        n = ctypes.c_uint32()
        self._check(lib.bbp_last_timings(self._h, arr, cap, ctypes.byref(n)))
        return [(int(arr[i]), float(arr[i + 1])) for i in range(0, n.value, 2)]


class Pool(Context):
    """A device pool (include/bbp.h "Device pool"): one handle, one engine context per GPU behind it.  Takes the host-pointer
    calls of Context -- prove / verify (combined and dealt to the least-loaded member), prove_batch / verify_batch /
    verify_batch_aggregated / msm_batch (block-split over the members, results in request order); the device-pointer calls
    need a member (`pool.member(i)`).  devices=None -> every visible GPU (bbp_init_all)."""

    def __init__(self, devices=None):
        self._devices = None if devices is None else list(devices)
        super().__init__(0)

    def _init(self, _device):
        if self._devices is None:
            return lib.bbp_init_all(ctypes.byref(self._h))
        arr = (ctypes.c_int32 * len(self._devices))(*self._devices)
        return lib.bbp_pool_init(arr, len(self._devices), ctypes.byref(self._h))

    def __len__(self):
        return lib.bbp_pool_size(self._h)

    def member(self, i):
        h = lib.bbp_pool_member(self._h, i)
        if not h:
            raise IndexError(i)
        return Context(_borrowed=h)

    def member_stats(self, i):
        """(combined prove / verify device calls dealt to member i, requests they carried)"""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(lib.bbp_pool_member_stats(self._h, i, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

"""ctypes loader for the C oracle (oracle/c/bbp_oracle.c). Test infrastructure / cpu_baseline only."""
import ctypes


class _Trace(ctypes.Structure):
    _fields_ = [("y", ctypes.c_uint8 * 32), ("z", ctypes.c_uint8 * 32), ("u", ctypes.c_uint8 * 32), ("x", ctypes.c_uint8 * 32),
                ("w", ctypes.c_uint8 * 32), ("u_ipp", (ctypes.c_uint8 * 32) * 16), ("n_mul", ctypes.c_int), ("n_cons", ctypes.c_int)]


class OracleC:
    def __init__(self, path):
        self.lib = L = ctypes.CDLL(path)
        vp, u32, i32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint64
        L.oc_prove.restype = i32
        L.oc_prove.argtypes = [vp, vp, u32, u64, vp, i32, i32, vp, ctypes.POINTER(u32), vp]
        L.oc_verify.restype = i32
        L.oc_verify.argtypes = [vp, u32, vp, vp, vp, vp, u32, i32, i32, vp]
        L.oc_msm_layout.restype = i32
        L.oc_msm_layout.argtypes = [vp, u32, u32, vp]
        L.oc_msm_layout_many.restype = i32
        L.oc_msm_layout_many.argtypes = [vp, vp, vp, i32, i32, vp]
        L.oc_prove_many.restype = i32
        L.oc_prove_many.argtypes = [vp, vp, u32, u32, i32, i32, i32, vp, u32, vp]
        L.oc_verify_many.restype = i32
        L.oc_verify_many.argtypes = [vp, u32, u32, u32, i32, i32, i32, vp]
        L.oc_generator.argtypes = [u32, vp]
        L.oc_mimc_constant.argtypes = [u32, vp]
        L.oc_witness.argtypes = [vp, i32, vp]
        L.oc_sc_op.argtypes = [i32, vp, vp, vp]
        L.oc_scalarmult.restype = i32
        L.oc_scalarmult.argtypes = [vp, vp, vp]
        L.oc_from_uniform.argtypes = [vp, vp]
        L.oc_merlin_kat.argtypes = [ctypes.c_char_p, ctypes.c_char_p, vp, u32, ctypes.c_char_p, vp, u32]

    @staticmethod
    def _b(x):
        return (ctypes.c_uint8 * len(x)).from_buffer_copy(bytes(x))

    def prove(self, scalars7, pub_list, toggle, entropy, rounds=90, cap=2048, want_trace=False):
        n = len(pub_list) // 32
        out = (ctypes.c_uint8 * (4096 + 32 * (4 + n)))()
        pl = ctypes.c_uint32()
        tr = _Trace()
        rc = self.lib.oc_prove(self._b(scalars7), self._b(pub_list), n, toggle, self._b(entropy), rounds, cap, out,
                               ctypes.byref(pl), ctypes.byref(tr))
        rec = bytes(out)[:pl.value + 32 * (4 + n)] if rc == 0 else b""
        if want_trace:
            trace = dict(y=bytes(tr.y).hex(), z=bytes(tr.z).hex(), u=bytes(tr.u).hex(), x=bytes(tr.x).hex(), w=bytes(tr.w).hex(),
                         u_ipp=[bytes(v).hex() for v in tr.u_ipp], n_mul=tr.n_mul, n_constraints=tr.n_cons)
            return rc, rec, trace
        return rc, rec

    def verify(self, record, score, z_img, seed, pub_list, rounds=90, cap=2048, entropy32=bytes(32)):
        n = len(pub_list) // 32
        return self.lib.oc_verify(self._b(record), len(record), self._b(score), self._b(z_img), self._b(seed),
                                  self._b(pub_list), n, rounds, cap, self._b(entropy32))

    def msm_layout(self, scalars, n, layout):
        out = (ctypes.c_uint8 * 32)()
        rc = self.lib.oc_msm_layout(self._b(scalars), n, layout, out)
        assert rc == 0
        return bytes(out)

    def msm_layout_many(self, rows, ns, layouts, threads):
        cnt = len(rows)
        out = (ctypes.c_uint8 * (32 * cnt))()
        rc = self.lib.oc_msm_layout_many(self._b(b"".join(rows)), (ctypes.c_uint32 * cnt)(*ns), (ctypes.c_uint32 * cnt)(*layouts),
                                         cnt, threads, out)
        assert rc == 0
        return bytes(out)

    def prove_many(self, inputs, entropy, B, N, threads, rounds=90, cap=2048):
        stride = 1121 + 32 * (4 + N)
        out = (ctypes.c_uint8 * (B * stride))()
        st = (ctypes.c_int * B)()
        self.lib.oc_prove_many(self._b(inputs), self._b(entropy), B, N, rounds, cap, threads, out, stride, st)
        return bytes(out), list(st)

    def verify_many(self, inputs, B, N, threads, rounds=90, cap=2048):
        rec_len = 1121 + 32 * (4 + N)
        st = (ctypes.c_int * B)()
        self.lib.oc_verify_many(self._b(inputs), B, N, rec_len, rounds, cap, threads, st)
        return list(st)

    def generator(self, i):
        out = (ctypes.c_uint8 * 32)()
        self.lib.oc_generator(i, out)
        return bytes(out)

    def mimc_constant(self, i):
        out = (ctypes.c_uint8 * 32)()
        self.lib.oc_mimc_constant(i, out)
        return bytes(out)

    def witness(self, dks, rounds=90):
        out = (ctypes.c_uint8 * 192)()
        self.lib.oc_witness(self._b(dks), rounds, out)
        return bytes(out)

    def sc_op(self, op, a, b=bytes(32)):
        out = (ctypes.c_uint8 * 32)()
        self.lib.oc_sc_op(op, self._b(a), self._b(b), out)
        return int.from_bytes(bytes(out), "little")

    def scalarmult(self, s, p):
        out = (ctypes.c_uint8 * 32)()
        return bytes(out) if self.lib.oc_scalarmult(self._b(s), self._b(p), out) else None

    def from_uniform(self, b64):
        out = (ctypes.c_uint8 * 32)()
        self.lib.oc_from_uniform(self._b(b64), out)
        return bytes(out)

    def merlin_kat(self, label, l1, m1, l2, n):
        out = (ctypes.c_uint8 * n)()
        self.lib.oc_merlin_kat(label, l1, self._b(m1) if m1 else None, len(m1), l2, out, n)
        return bytes(out)


_cache = {}


def load(path):
    if path not in _cache:
        _cache[path] = OracleC(path)
    return _cache[path]

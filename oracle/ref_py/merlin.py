"""ORACLE (test infrastructure, never shipped): merlin 1.3.0 transcript over STROBE-128 / Keccak-f[1600].

Follows merlin 1.3.0 (Cargo.lock:399-401; crate source not under /root/reference) as restated in
SURVEY.md App. A.1; call site src/blindbid/mod.rs:37 (`Transcript::new(b"BlindBidProofGadget")`).
Pinned by merlin's published "test protocol" vector (tests/test_oracle_kat.py).
"""
import struct

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def keccak_f1600(state: bytearray):
    a = list(struct.unpack("<25Q", state))
    A = [[a[x + 5 * y] for y in range(5)] for x in range(5)]
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        Dv = [C[(x - 1) % 5] ^ (((C[(x + 1) % 5] << 1) | (C[(x + 1) % 5] >> 63)) & _M) for x in range(5)]
        A = [[A[x][y] ^ Dv[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                r = _ROT[x][y]
                v = A[x][y]
                B[y][(2 * x + 3 * y) % 5] = ((v << r) | (v >> (64 - r))) & _M if r else v
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= rc
    state[:] = struct.pack("<25Q", *[A[i % 5][i // 5] for i in range(25)])


_R = 166
_FI, _FA, _FC, _FT, _FM, _FK = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label: bytes = b"", _clone=None):
        if _clone is not None:
            self.st = bytearray(_clone.st)
            self.pos, self.pos_begin, self.cur_flags = _clone.pos, _clone.pos_begin, _clone.cur_flags
            return
        self.st = bytearray(200)
        self.st[0:6] = bytes([1, _R + 2, 1, 0, 1, 96])
        self.st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(self.st)
        self.pos = self.pos_begin = self.cur_flags = 0
        self.meta_ad(protocol_label, False)

    def clone(self):
        return Strobe128(_clone=self)

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[_R + 1] ^= 0x80
        keccak_f1600(self.st)
        self.pos = self.pos_begin = 0

    def _absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == _R:
                self._run_f()

    def _overwrite(self, data):
        for b in data:
            self.st[self.pos] = b
            self.pos += 1
            if self.pos == _R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray(n)
        for i in range(n):
            out[i] = self.st[self.pos]
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == _R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & _FT == 0
        old = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old, flags]))
        if flags & (_FC | _FK) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(_FM | _FA, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(_FA, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(_FI | _FA | _FC, more)
        return self._squeeze(n)

    def key(self, data, more):
        self._begin_op(_FA | _FC, more)
        self._overwrite(data)


class Transcript:
    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, msg):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", len(msg)), True)
        self.strobe.ad(msg, False)

    def append_u64(self, label, x):
        self.append_message(label, struct.pack("<Q", x))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", n), True)
        return self.strobe.prf(n, False)

    def build_rng(self, witnesses, entropy32):
        """TranscriptRngBuilder: rekey_with_witness_bytes(label, w) for each, then finalize(rng)
        where `entropy32` replaces the 32 bytes drawn from the external rng."""
        s = self.strobe.clone()
        for label, w in witnesses:
            s.meta_ad(label, False)
            s.meta_ad(struct.pack("<I", len(w)), True)
            s.key(w, False)
        assert len(entropy32) == 32
        s.meta_ad(b"rng", False)
        s.key(entropy32, False)
        return TranscriptRng(s)


class TranscriptRng:
    def __init__(self, strobe):
        self.strobe = strobe

    def fill_bytes(self, n):
        self.strobe.meta_ad(struct.pack("<I", n), False)
        return self.strobe.prf(n, False)

"""Test-side client of the UDS server: an independent Python restatement of the framing (server/tlv.h is the C++ one; both follow
the layout assumed there -- width byte, little-endian length, payload) and of the request / response layouts of the reference
(src/futures/main.rs:70-101, src/blindbid/proof.rs:97-143, verify.rs:91-128)."""
import socket


def tlv(payload):
    n = len(payload)
    width = 1 if n <= 0xff else 2 if n <= 0xffff else 4 if n <= 0xffffffff else 8
    return bytes([width]) + n.to_bytes(width, "little") + bytes(payload)


def tlv_list(items):
    return tlv(b"".join(tlv(i) for i in items))


def parse(buf, pos=0):
    """-> (payload, next position)"""
    width = buf[pos]
    assert width in (1, 2, 4, 8), width
    n = int.from_bytes(buf[pos + 1:pos + 1 + width], "little")
    start = pos + 1 + width
    assert start + n <= len(buf)
    return buf[start:start + n], start + n


def parse_list(buf, pos=0):
    inner, nxt = parse(buf, pos)
    items, p = [], 0
    while p < len(inner):
        it, p = parse(inner, p)
        items.append(it)
    return items, nxt


def prove_request(scalars7, pub_list, toggle):
    """opcode 1 || S(d..seed) x 7 || LIST(bids) || U64(toggle), as one frame"""
    n = len(pub_list) // 32
    body = b"\x01" + b"".join(tlv(scalars7[32 * i:32 * i + 32]) for i in range(7))
    body += tlv_list([pub_list[32 * i:32 * i + 32] for i in range(n)]) + tlv(int(toggle).to_bytes(8, "little"))
    return tlv(body)


def decode_proof(blob):
    """TLV(proof) || LIST(commitments) || LIST(t_c)  ->  (proof bytes, [commitments], [t_c])"""
    proof, p = parse(blob, 0)
    c, p = parse_list(blob, p)
    t, p = parse_list(blob, p)
    assert p == len(blob)
    return proof, c, t


def verify_request(proof_blob, score, z_img, seed, pub_list):
    n = len(pub_list) // 32
    body = b"\x02" + tlv(proof_blob) + tlv(score) + tlv(z_img) + tlv(seed) + tlv_list([pub_list[32 * i:32 * i + 32] for i in range(n)])
    return tlv(body)


class Conn:
    def __init__(self, path, timeout=60.0):
        self.s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self.s.settimeout(timeout)
        self.s.connect(path)

    def send(self, data):
        self.s.sendall(data)

    def _read(self, n):
        out = b""
        while len(out) < n:
            try:
                c = self.s.recv(n - len(out))
            except ConnectionResetError:  # the server dropped the socket with part of our request unread: also "no payload"
                return None
            if not c:
                return None
            out += c
        return out

    def recv_frame(self):
        """payload of the next frame, or None when the server closed the connection without writing one"""
        w = self._read(1)
        if w is None:
            return None
        ln = self._read(w[0])
        if ln is None:
            return None
        return self._read(int.from_bytes(ln, "little"))

    def close(self):
        self.s.close()


def prove(path, scalars7, pub_list, toggle):
    c = Conn(path)
    try:
        c.send(prove_request(scalars7, pub_list, toggle))
        return c.recv_frame()
    finally:
        c.close()


def verify(path, proof_blob, score, z_img, seed, pub_list):
    c = Conn(path)
    try:
        c.send(verify_request(proof_blob, score, z_img, seed, pub_list))
        return c.recv_frame()
    finally:
        c.close()

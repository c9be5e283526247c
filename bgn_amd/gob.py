"""Go `encoding/gob` envelopes of the reference's ciphertexts (ciphertext.go:76-116, bgn.go:501-560).

`Ciphertext.Bytes()` and `PolyCiphertext.Bytes()` gob-encode two small wrapper structs around
`Element.Bytes()`:

    type ciphertextWrapper     struct { CBytes []byte; L2 bool }                                  // ciphertext.go:17-20
    type polyCiphertextWrapper struct { CoeffBytes [][]byte; Degree, ScaleFactor int; L2 bool }   // ciphertext.go:33-38

This module writes and reads exactly those streams so arrays of marshalled ciphertexts produced by
existing Go services can be unpacked into the engine's dense wire arrays (`unpack_ciphertexts`) and
results handed back in the form `NewCiphertextFromBytes` expects.  It is host-side format code: no
arithmetic, no GPU.

Stream grammar (encoding/gob package documentation, "Encoding Details"):
  * unsigned integer: one byte if < 128, else the negated byte count in one byte followed by the
    big-endian bytes; signed integer i: (i << 1), or (^i << 1) | 1 when negative, sent as unsigned;
    bool: unsigned 0/1; []byte and string: unsigned length + bytes; slice: unsigned count + elements;
  * struct: (field-number delta, value) pairs in field order, zero-valued fields omitted, terminated by a
    0 delta; field numbering starts at -1;
  * a stream is a sequence of messages, each an unsigned byte count followed by a signed type id:
    negative = definition of type -id (a `wireType` value follows), positive = a value of that type;
    user type ids start at 65 in a fresh encoder and a struct's definition precedes those of its element
    types.  A Go process assigns ids in the order it first meets types, so a *reader* must take the ids
    and field order from the definitions in the stream, as this one does; a *writer* may use 65, 66.
The known-answer vector in the package documentation (type Point struct{X, Y int}; Point{22, 33}) pins
both directions in tests/test_gob_envelope.py.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

# predefined type ids (encoding/gob type.go)
T_BOOL, T_INT, T_UINT, T_FLOAT, T_BYTES, T_STRING = 1, 2, 3, 4, 5, 6
FIRST_USER_ID = 65


class GobError(ValueError):
    pass


# ----------------------------------------------------------------------------- primitives
def enc_uint(v: int) -> bytes:
    if v < 0:
        raise GobError("negative value for an unsigned field")
    if v < 128:
        return bytes([v])
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def enc_int(v: int) -> bytes:
    return enc_uint(((~v) << 1) | 1 if v < 0 else v << 1)


def enc_bytes(b: bytes) -> bytes:
    return enc_uint(len(b)) + bytes(b)


class _Reader:
    def __init__(self, data: bytes, pos: int = 0, end: Optional[int] = None):
        self.d, self.p, self.e = data, pos, len(data) if end is None else end

    def take(self, n: int) -> bytes:
        if n < 0 or self.p + n > self.e:
            raise GobError("truncated gob stream")
        out = self.d[self.p:self.p + n]
        self.p += n
        return out

    def uint(self) -> int:
        b = self.take(1)[0]
        if b < 128:
            return b
        n = 256 - b
        if n > 8:
            raise GobError("unsigned integer wider than 64 bits")
        return int.from_bytes(self.take(n), "big")

    def int(self) -> int:
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def bytes_(self) -> bytes:
        return bytes(self.take(self.uint()))

    def done(self) -> bool:
        return self.p >= self.e


# ----------------------------------------------------------------------------- type definitions
# A user type is ("struct", name, [(field name, type id), ...]) or ("slice", name, element type id).
def _enc_common(name: str, tid: int) -> bytes:
    # CommonType{Name string; Id typeId}
    return b"\x01" + enc_bytes(name.encode()) + b"\x01" + enc_int(tid) + b"\x00"


def enc_struct_def(tid: int, name: str, fields: Sequence[Tuple[str, int]]) -> bytes:
    """Message defining struct type `tid`: wireType{StructT: &structType{CommonType, Field []*fieldType}}."""
    body = enc_int(-tid) + b"\x03" + b"\x01" + _enc_common(name, tid)
    if fields:
        body += b"\x01" + enc_uint(len(fields))
        for fname, ftid in fields:
            body += b"\x01" + enc_bytes(fname.encode()) + b"\x01" + enc_int(ftid) + b"\x00"
    body += b"\x00\x00"
    return enc_uint(len(body)) + body


def enc_slice_def(tid: int, name: str, elem: int) -> bytes:
    """Message defining slice type `tid`: wireType{SliceT: &sliceType{CommonType, Elem typeId}}."""
    body = enc_int(-tid) + b"\x02" + b"\x01" + _enc_common(name, tid) + b"\x01" + enc_int(elem) + b"\x00\x00"
    return enc_uint(len(body)) + body


def _dec_common(r: _Reader) -> Tuple[str, int]:
    name, tid, field = "", 0, -1
    while True:
        d = r.uint()
        if d == 0:
            return name, tid
        field += d
        if field == 0:
            name = r.bytes_().decode("utf-8", "replace")
        elif field == 1:
            tid = r.int()
        else:
            raise GobError("unknown CommonType field")


def _dec_wire_type(r: _Reader):
    """wireType: fields ArrayT(0) SliceT(1) StructT(2) MapT(3) GobEncoderT(4) BinaryMarshalerT(5) TextMarshalerT(6)."""
    out, field = None, -1
    while True:
        d = r.uint()
        if d == 0:
            break
        field += d
        if field == 1:                                   # sliceType{CommonType; Elem}
            name, elem, f = "", 0, -1
            while True:
                dd = r.uint()
                if dd == 0:
                    break
                f += dd
                if f == 0:
                    name, _ = _dec_common(r)
                elif f == 1:
                    elem = r.int()
                else:
                    raise GobError("unknown sliceType field")
            out = ("slice", name, elem)
        elif field == 2:                                 # structType{CommonType; Field []*fieldType}
            name, fields, f = "", [], -1
            while True:
                dd = r.uint()
                if dd == 0:
                    break
                f += dd
                if f == 0:
                    name, _ = _dec_common(r)
                elif f == 1:
                    for _ in range(r.uint()):
                        fields.append(_dec_common(r))    # fieldType has the same shape: {Name string; Id typeId}
                else:
                    raise GobError("unknown structType field")
            out = ("struct", name, fields)
        else:
            raise GobError("gob type kind %d is not used by the ciphertext envelopes" % field)
    if out is None:
        raise GobError("empty type definition")
    return out


# ----------------------------------------------------------------------------- values
def _enc_value(types: Dict[int, Any], tid: int, v) -> Optional[bytes]:
    """Encoding of v as type tid, or None when v is the zero value (struct fields omit those)."""
    if tid == T_BOOL:
        return b"\x01" if v else None
    if tid == T_INT:
        return enc_int(int(v)) if v else None
    if tid == T_UINT:
        return enc_uint(int(v)) if v else None
    if tid in (T_BYTES, T_STRING):
        b = v.encode() if isinstance(v, str) else bytes(v)
        return enc_bytes(b) if b else None
    kind = types[tid]
    if kind[0] == "slice":
        if not v:
            return None
        out = enc_uint(len(v))
        for e in v:
            ev = _enc_value(types, kind[2], e)
            if ev is None:                               # elements are always sent, zero or not
                ev = {T_BOOL: b"\x00", T_INT: b"\x00", T_UINT: b"\x00", T_BYTES: b"\x00", T_STRING: b"\x00"}[kind[2]]
            out += ev
        return out
    out, last = b"", -1
    for i, (fname, ftid) in enumerate(kind[2]):
        ev = _enc_value(types, ftid, v.get(fname))
        if ev is not None:
            out += enc_uint(i - last) + ev
            last = i
    return out + b"\x00"


def _dec_value(types: Dict[int, Any], tid: int, r: _Reader):
    if tid == T_BOOL:
        return r.uint() != 0
    if tid == T_INT:
        return r.int()
    if tid == T_UINT:
        return r.uint()
    if tid == T_BYTES:
        return r.bytes_()
    if tid == T_STRING:
        return r.bytes_().decode("utf-8", "replace")
    if tid not in types:
        raise GobError("value of undefined type %d" % tid)
    kind = types[tid]
    if kind[0] == "slice":
        return [_dec_value(types, kind[2], r) for _ in range(r.uint())]
    out, field = {}, -1
    while True:
        d = r.uint()
        if d == 0:
            return out
        field += d
        if field >= len(kind[2]):
            raise GobError("field number out of range")
        fname, ftid = kind[2][field]
        out[fname] = _dec_value(types, ftid, r)


def encode_struct(name: str, fields: Sequence[Tuple[str, int]], value: Dict[str, Any],
                  extra_types: Sequence[Tuple[int, str, int]] = ()) -> bytes:
    """A complete stream as a fresh gob.Encoder writes it for one struct value: the struct's definition
    (id 65), the definitions of its slice-typed fields (`extra_types`: (id, name, element id)), the value."""
    types: Dict[int, Any] = {FIRST_USER_ID: ("struct", name, list(fields))}
    out = enc_struct_def(FIRST_USER_ID, name, fields)
    for tid, tname, elem in extra_types:
        types[tid] = ("slice", tname, elem)
        out += enc_slice_def(tid, tname, elem)
    body = enc_int(FIRST_USER_ID) + _enc_value(types, FIRST_USER_ID, value)
    return out + enc_uint(len(body)) + body


def decode_struct(data: bytes) -> Tuple[str, Dict[str, Any]]:
    """First struct value of a gob stream -> (type name, {field name: value}); absent fields are zero
    values and simply missing from the dict (as gob.Decoder leaves them untouched)."""
    if not data:
        raise GobError("no data provided")               # bgn.go:503-505
    r = _Reader(bytes(data))
    types: Dict[int, Any] = {}
    while not r.done():
        n = r.uint()
        m = _Reader(r.d, r.p, r.p + n)
        r.take(n)
        tid = m.int()
        if tid < 0:
            types[-tid] = _dec_wire_type(m)
            continue
        if tid not in types or types[tid][0] != "struct":
            raise GobError("top-level value is not a struct")
        return types[tid][1], _dec_value(types, tid, m)
    raise GobError("gob stream holds no value")


# ----------------------------------------------------------------------------- the two envelopes
_CT_FIELDS = [("CBytes", T_BYTES), ("L2", T_BOOL)]
_POLY_FIELDS = [("CoeffBytes", FIRST_USER_ID + 1), ("Degree", T_INT), ("ScaleFactor", T_INT), ("L2", T_BOOL)]


def marshal_ciphertext(c_bytes: bytes, l2: bool) -> bytes:
    """Ciphertext.Bytes(), ciphertext.go:76-92."""
    return encode_struct("ciphertextWrapper", _CT_FIELDS, {"CBytes": c_bytes, "L2": l2})


def unmarshal_ciphertext(data: bytes) -> Tuple[bytes, bool]:
    """The gob half of NewCiphertextFromBytes, bgn.go:501-526 -> (CBytes, L2)."""
    _, v = decode_struct(data)
    return bytes(v.get("CBytes", b"")), bool(v.get("L2", False))


def marshal_poly_ciphertext(coeff_bytes: Sequence[bytes], degree: int, scale: int, l2: bool) -> bytes:
    """PolyCiphertext.Bytes(), ciphertext.go:94-116."""
    return encode_struct("polyCiphertextWrapper", _POLY_FIELDS,
                         {"CoeffBytes": [bytes(b) for b in coeff_bytes], "Degree": degree, "ScaleFactor": scale, "L2": l2},
                         extra_types=[(FIRST_USER_ID + 1, "[][]uint8", T_BYTES)])


def unmarshal_poly_ciphertext(data: bytes) -> Tuple[List[bytes], int, int, bool]:
    """The gob half of NewPolyCiphertextFromBytes, bgn.go:530-560 -> (CoeffBytes, Degree, ScaleFactor, L2)."""
    _, v = decode_struct(data)
    return ([bytes(b) for b in v.get("CoeffBytes", [])], int(v.get("Degree", 0)), int(v.get("ScaleFactor", 0)),
            bool(v.get("L2", False)))


def unpack_ciphertexts(blobs: Sequence[bytes], elem_bytes: int) -> Tuple[bytes, List[bool]]:
    """Marshalled ciphertexts -> one dense wire array for the batch entry points + the level flags.
    An empty CBytes (gob omits zero-length slices) is the all-zero identity encoding."""
    parts, levels = [], []
    for blob in blobs:
        c, l2 = unmarshal_ciphertext(blob)
        if not c:
            c = bytes(elem_bytes)
        if len(c) != elem_bytes:
            raise GobError("element of %d bytes, expected %d" % (len(c), elem_bytes))
        parts.append(c)
        levels.append(l2)
    return b"".join(parts), levels

"""CPU: the gob envelopes of Ciphertext.Bytes / PolyCiphertext.Bytes (ciphertext.go:76-116, bgn.go:501-560;
SURVEY.md section 8(f) rank 1).  The format is Go's encoding/gob (standard library, not in the reference tree);
it is pinned by the known-answer vectors printed in the encoding/gob package documentation."""
import pytest

from bgn_amd import gob

# encoding/gob package documentation, "Encoding Details":
#   type Point struct { X, Y int };  the stream for Point{22, 33}
DOC_POINT = bytes.fromhex(
    "1f ff 81 03 01 01 05 50 6f 69 6e 74 01 ff 82 00 01 02 01 01 58 01 04 00 01 01 59 01 04 00 00 00"
    "07 ff 82 01 2c 01 42 00".replace(" ", ""))


def test_documented_integer_encodings():
    assert gob.enc_uint(7) == b"\x07"                       # "7 is transmitted as 07"
    assert gob.enc_uint(256) == bytes.fromhex("fe0100")     # "256 is transmitted as (FE 01 00)"
    assert gob.enc_int(-129) == bytes.fromhex("fe0101")     # "-129 ... (^(-129) << 1) | 1 = 257: (FE 01 01)"
    for v in [0, 1, 127, 128, 255, 256, 65535, 2 ** 40, 2 ** 63 - 1]:
        r = gob._Reader(gob.enc_uint(v))
        assert r.uint() == v and r.done()
    for v in [0, 1, -1, 63, -64, 64, -65, 10 ** 12, -10 ** 12]:
        r = gob._Reader(gob.enc_int(v))
        assert r.int() == v and r.done()


def test_documented_struct_stream_both_directions():
    fields = [("X", gob.T_INT), ("Y", gob.T_INT)]
    assert gob.encode_struct("Point", fields, {"X": 22, "Y": 33}) == DOC_POINT
    assert gob.decode_struct(DOC_POINT) == ("Point", {"X": 22, "Y": 33})


def test_ciphertext_envelope_round_trip_and_layout():
    c = bytes(range(1, 37))
    for l2 in (False, True):
        blob = gob.marshal_ciphertext(c, l2)
        assert gob.unmarshal_ciphertext(blob) == (c, l2)
    # layout of the value message: id 65, field 1 = CBytes, field 2 = L2 only when true
    blob = gob.marshal_ciphertext(c, True)
    assert blob.endswith(b"\xff\x82\x01" + bytes([len(c)]) + c + b"\x01\x01\x00")
    assert gob.marshal_ciphertext(c, False).endswith(b"\xff\x82\x01" + bytes([len(c)]) + c + b"\x00")
    assert b"ciphertextWrapper" in blob and b"CBytes" in blob and b"L2" in blob
    with pytest.raises(gob.GobError, match="no data provided"):      # bgn.go:503-505
        gob.unmarshal_ciphertext(b"")
    with pytest.raises(gob.GobError):
        gob.unmarshal_ciphertext(blob[:-3])


def test_reader_takes_ids_and_fields_from_the_stream():
    """A Go process that has encoded other types first assigns later ids; the reader must not assume 65."""
    tid = 71
    types = {tid: ("struct", "ciphertextWrapper", [("CBytes", gob.T_BYTES), ("L2", gob.T_BOOL)])}
    body = gob.enc_int(tid) + gob._enc_value(types, tid, {"CBytes": b"\x05\x06", "L2": True})
    stream = gob.enc_struct_def(tid, "ciphertextWrapper", types[tid][2]) + gob.enc_uint(len(body)) + body
    assert gob.unmarshal_ciphertext(stream) == (b"\x05\x06", True)
    # an older sender without the L2 field: the receiver keeps the zero value
    types = {tid: ("struct", "ciphertextWrapper", [("CBytes", gob.T_BYTES)])}
    body = gob.enc_int(tid) + gob._enc_value(types, tid, {"CBytes": b"\x09"})
    stream = gob.enc_struct_def(tid, "ciphertextWrapper", types[tid][2]) + gob.enc_uint(len(body)) + body
    assert gob.unmarshal_ciphertext(stream) == (b"\x09", False)


def test_poly_envelope_round_trip():
    coeffs = [bytes([i]) * 20 for i in range(1, 6)]
    for degree, scale, l2 in [(5, 0, False), (5, 3, True), (0, 0, False), (5, -2, True)]:
        blob = gob.marshal_poly_ciphertext(coeffs, degree, scale, l2)
        assert gob.unmarshal_poly_ciphertext(blob) == (coeffs, degree, scale, l2)
    blob = gob.marshal_poly_ciphertext(coeffs, 5, 3, True)
    assert b"polyCiphertextWrapper" in blob and b"CoeffBytes" in blob and b"[][]uint8" in blob
    # struct definition (id 65) first, then the [][]byte slice type (id 66), then the value
    assert blob.index(b"polyCiphertextWrapper") < blob.index(b"[][]uint8")
    assert gob.unmarshal_poly_ciphertext(gob.marshal_poly_ciphertext([], 0, 0, False)) == ([], 0, 0, False)


def test_unpack_marshalled_ciphertexts_into_a_dense_wire_array():
    E = 36
    elems = [bytes([i]) * E for i in range(1, 5)] + [bytes(E)]
    blobs = [gob.marshal_ciphertext(e, i % 2 == 1) for i, e in enumerate(elems)]
    wire, levels = gob.unpack_ciphertexts(blobs, E)
    assert wire == b"".join(elems) and levels == [False, True, False, True, False]
    with pytest.raises(gob.GobError):
        gob.unpack_ciphertexts([gob.marshal_ciphertext(b"\x01" * 5, False)], E)


def test_mirror_objects_marshal_like_the_reference_api():
    from bgn_amd.api import Ciphertext, PolyCiphertext
    c = Ciphertext(bytes(range(40)), True)
    assert gob.unmarshal_ciphertext(c.Bytes()) == (c.C, True)
    pc = PolyCiphertext([Ciphertext(bytes([7]) * 40, False), Ciphertext(bytes([9]) * 40, False)], 2, 1, False)
    assert gob.unmarshal_poly_ciphertext(pc.Bytes()) == ([bytes([7]) * 40, bytes([9]) * 40], 2, 1, False)


@pytest.mark.gpu
def test_gpu_marshalled_ciphertexts_survive_the_round_trip():
    """bgn_test.go:37-85: Bytes() -> New...FromBytes preserves the element; here additionally the batch path
    consumes an array of marshalled ciphertexts directly."""
    from conftest import engine_key, load_fixture
    fx = load_fixture("k256")
    pk, sk = engine_key(fx)
    pk.SetupDecryption(sk)
    cts = pk.EncryptBatch([3, 0, 7, 11], [5, 0, 9, 13])            # the second one is the identity
    back = [pk.NewCiphertextFromBytes(c.Bytes()) for c in cts]
    assert [b.C for b in back] == [c.C for c in cts] and not any(b.L2 for b in back)
    l2 = pk.Mult(cts[0], cts[2])
    assert pk.NewCiphertextFromBytes(l2.Bytes()) == l2
    poly = pk.EncryptPoly([1, -1, 0, 1], scale=2)
    assert pk.NewPolyCiphertextFromBytes(poly.Bytes()) == poly
    wire, levels = gob.unpack_ciphertexts([c.Bytes() for c in cts], pk.engine.elem_bytes)
    m, st = pk.engine.decrypt(1, wire)
    assert not st.any() and [int(v) for v in m] == [3, 0, 7, 11] and levels == [False] * 4

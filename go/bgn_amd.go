// bgn_amd.go — cgo binding of the MI355X batch engine (include/bgn_amd.h) for package bgn of sachaservan/bgn.
//
// NOT COMPILED HERE: the image this repository is built and tested in has no Go toolchain (`go: command not found`,
// also on the GPU box) and no libpbc, so this file has never been through `go vet` or `go build`.  The C ABI it binds is
// exercised end to end from C99 in the cgo call shape (tests/cpp/cgo_shape.c), from C++ (include/bgn_amd.hpp) and from
// Python ctypes (bgn_amd/_lib.py); every C call below has its twin there.  Build lines: go/README.md.
//
// Drop the file into the reference's package directory.  It adds:
//   - Engine (one bgn_ctx: a PublicKey's pairing context on one GPU) and the batch methods EncryptBatch, AddBatch,
//     SubBatch, NegBatch, MultBatch, MakeL2Batch, MultConstBatch, DecryptBatch, MultPolyBatch, MultConstPolyOne,
//     EvalPolyBatch, CheckDecryptionProofBatch, CheckPlaintextKnowledgeBatch, ValidateBatch, Calibrate;
//   - DeviceArray and the *Dev methods (EncryptBatchDev, AddBatchDev, SubBatchDev, NegBatchDev, MultBatchDev,
//     MakeL2BatchDev, MultConstBatchDev, MultPolyBatchDev, DecryptBatchDev, ValidateBatchDev): arrays that stay on the
//     GPU between calls, for chains like MultPoly -> AddPoly -> Decrypt;
//   - the bodies of the reference's single-element methods as count-1 calls (engineMult, engineAdd, engineSub,
//     engineNeg, engineMakeL2, engineMultConst, engineEncryptWithRandomness, engineDecrypt): to switch a method over,
//     make its body in bgn.go `return pk.engineMult(ct1, ct2)` and so on — signatures and panics stay as they are;
//     concurrent count-1 calls are merged inside the library (include/bgn_amd.h, "Concurrent small calls"), so the
//     goroutine-per-pair loops of poly.go:97-109 and :139-153 fill the GPU unchanged;
//   - MultiEngine (one bgn_mctx: the same key on several GPUs of the node, batches sharded by element, MultPoly by
//     polynomial).
//
// Reference lines each method replaces are cited at the method.
package bgn

/*
#include <stdint.h>
#include <stdlib.h>
#include "bgn_amd.h"
*/
import "C"

import (
	"encoding/binary"
	"errors"
	"math/big"
	"runtime"
	"sync"
	"unsafe"

	"github.com/Nik-U/pbc"
)

// ---- helpers ----------------------------------------------------------------------------------------------------------

var errNoDL = errors.New("cannot find discrete log; out of bounds") // gsbs.go:105

func engineErr(rc C.int) error {
	if rc == 0 {
		return nil
	}
	return errors.New(C.GoString(C.bgn_last_error()))
}

// locked runs one engine call and, when it fails, fetches its message on the SAME operating-system thread:
// bgn_last_error() is thread-local on the C side, and the Go scheduler may move a goroutine to another thread
// between two cgo calls (the message would then be empty, or another call's).  Every call below goes through here.
func locked(call func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	return engineErr(call())
}

// u8 is the address of a byte slice for the C side; nil for an absent (nil or empty) array.
func u8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

func allZero(b []byte) bool {
	for _, v := range b {
		if v != 0 {
			return false
		}
	}
	return true
}

// leftPad returns b as exactly n big-endian bytes (b must not be longer).
func leftPad(b []byte, n int) []byte {
	if len(b) > n {
		panic("leftPad: value does not fit")
	}
	out := make([]byte, n)
	copy(out[n-len(b):], b)
	return out
}

// flatten packs big-endian scalars into one array of `width` bytes each.
func flatten(vals [][]byte, width int) []byte {
	out := make([]byte, 0, len(vals)*width)
	for _, v := range vals {
		out = append(out, leftPad(v, width)...)
	}
	return out
}

// scalars packs non-negative big integers as big-endian values of one common width; returns the array and the width.
func scalars(vals []*big.Int) ([]byte, int) {
	width := 1
	for _, v := range vals {
		if v.Sign() < 0 {
			panic("scalars: the ABI takes unsigned scalars (reduce modulo N first)")
		}
		if n := (v.BitLen() + 7) / 8; n > width {
			width = n
		}
	}
	out := make([]byte, 0, len(vals)*width)
	for _, v := range vals {
		out = append(out, leftPad(v.Bytes(), width)...)
	}
	return out, width
}

// ---- Engine -------------------------------------------------------------------------------------------------------------

// Engine wraps one bgn_ctx: the pairing context of a PublicKey on one GPU.
type Engine struct {
	h  *C.bgn_ctx
	L  int // bytes per F_p value; an element is 2L bytes (pbc Element.Bytes())
	pk *PublicKey
}

// engines maps a PublicKey to the engine the single-element methods use (a struct field cannot be added from here).
var engines sync.Map // *PublicKey -> *Engine

// NewEngine replaces pbc.NewPairing for the batch path (bgn.go:101, :640).
// The A1 parameters come from pk.PairingParams ("type a1 / p / n / l", bgn.go:583-593).
func (pk *PublicKey) NewEngine(device int) (*Engine, error) {
	p, l, err := parseA1(pk.PairingParams)
	if err != nil {
		return nil, err
	}
	pb, nb := p.Bytes(), pk.N.Bytes()
	P, Q := pk.P.Bytes(), pk.Q.Bytes() // PBC wire format, bgn.go:605-607
	det := 0
	if pk.Deterministic {
		det = 1
	}
	var h *C.bgn_ctx
	err = locked(func() C.int {
		return C.bgn_ctx_create(&h, u8(pb), C.size_t(len(pb)), u8(nb), C.size_t(len(nb)), C.uint64_t(l), u8(P), u8(Q),
		C.int(det), C.int(device))
	})
	if err != nil {
		return nil, err
	}
	e := &Engine{h: h, L: int(C.bgn_fp_bytes(h)), pk: pk}
	engines.Store(pk, e)
	return e, nil
}

// parseA1 reads p and l from the PBC parameter string (the reference's parseLFromPBCParams reads l, bgn.go:583-593).
func parseA1(params string) (*big.Int, uint64, error) {
	var p, l *big.Int
	fields := splitFields(params)
	for i := 0; i+1 < len(fields); i++ {
		switch fields[i] {
		case "p":
			p, _ = new(big.Int).SetString(fields[i+1], 10)
		case "l":
			l, _ = new(big.Int).SetString(fields[i+1], 10)
		}
	}
	if p == nil || l == nil || !l.IsUint64() {
		return nil, 0, errors.New("bgn_amd: cannot read p / l from the pairing parameters")
	}
	return p, l.Uint64(), nil
}

func splitFields(s string) []string {
	var out []string
	cur := ""
	for _, r := range s {
		if r == ' ' || r == '\n' || r == '\t' || r == '\r' {
			if cur != "" {
				out = append(out, cur)
				cur = ""
			}
		} else {
			cur += string(r)
		}
	}
	if cur != "" {
		out = append(out, cur)
	}
	return out
}

func (e *Engine) Close() {
	engines.Delete(e.pk)
	C.bgn_ctx_destroy(e.h)
}

func (pk *PublicKey) engine() *Engine {
	v, ok := engines.Load(pk)
	if !ok {
		panic("bgn_amd: no engine for this public key (call pk.NewEngine first)")
	}
	return v.(*Engine)
}

// SetMemoryBudget bounds the device memory of this key's tables and workspace (0: no bound).
func (e *Engine) SetMemoryBudget(bytes uint64) error {
	return locked(func() C.int { return C.bgn_ctx_set_memory_budget(e.h, C.uint64_t(bytes)) })
}

// MemoryBytes is the device memory the context holds now.
func (e *Engine) MemoryBytes() uint64 { return uint64(C.bgn_ctx_memory_bytes(e.h)) }

// SetOption sets a named knob of the context (include/bgn_amd.h "Options").
func (e *Engine) SetOption(name string, value int64) error {
	cs := C.CString(name)
	defer C.free(unsafe.Pointer(cs))
	return locked(func() C.int { return C.bgn_ctx_set_option(e.h, cs, C.int64_t(value)) })
}

// SetupDecryption replaces pk.SetupDecryption / ComputeDecryptionPreprocessing (bgn.go:195-201, :142-149).
func (e *Engine) SetupDecryption(sk *SecretKey) error {
	q1 := sk.Key.Bytes()
	if err := locked(func() C.int { return C.bgn_ctx_set_secret(e.h, u8(q1), C.size_t(len(q1))) }); err != nil {
		return err
	}
	if !e.pk.MsgSpace.IsUint64() {
		return errors.New("bgn_amd: message space beyond 64 bits")
	}
	return locked(func() C.int { return C.bgn_ctx_setup_decryption(e.h, C.uint64_t(e.pk.MsgSpace.Uint64())) })
}

// wire is Element.Bytes() of one ciphertext; the G1 identity (encryptZero(), bgn.go:562) is 2L zero bytes.
func (e *Engine) wire(ct *Ciphertext) []byte {
	if !ct.L2 && ct.C.Is0() {
		return make([]byte, 2*e.L)
	}
	return ct.C.Bytes()
}

// pack concatenates the wire bytes of level-uniform ciphertexts.
func (e *Engine) pack(cts []*Ciphertext) []byte {
	buf := make([]byte, 0, len(cts)*2*e.L)
	for _, ct := range cts {
		buf = append(buf, e.wire(ct)...)
	}
	return buf
}

func (e *Engine) unpackOne(b []byte, l2 bool) *Ciphertext {
	var el *pbc.Element
	if l2 {
		el = e.pk.Pairing.NewGT().NewFieldElement()
		el.SetBytes(b)
	} else {
		el = e.pk.G1.NewFieldElement() // identity
		if !allZero(b) {
			el.SetBytes(b)
		}
	}
	return &Ciphertext{el, l2}
}

func (e *Engine) unpack(buf []byte, l2 bool) []*Ciphertext {
	out := make([]*Ciphertext, len(buf)/(2*e.L))
	for i := range out {
		out[i] = e.unpackOne(buf[i*2*e.L:(i+1)*2*e.L], l2)
	}
	return out
}

func levelOf(l2 bool) C.int {
	if l2 {
		return 2
	}
	return 1
}

// randomness packs caller-supplied blinding scalars r[i] < N (nil: deterministic); the width is that of N.
func (e *Engine) randomness(r []*big.Int) ([]byte, int) {
	if r == nil {
		return nil, 0
	}
	w := len(e.pk.N.Bytes())
	out := make([]byte, 0, len(r)*w)
	for _, v := range r {
		out = append(out, leftPad(v.Bytes(), w)...)
	}
	return out, w
}

// ---- batch methods ------------------------------------------------------------------------------------------------------

// EncryptBatch is pk.EncryptWithRandomness / EncryptDeterministic (bgn.go:340-353, :325-331) over arrays;
// r == nil gives the deterministic form.  Negative plaintexts are reduced modulo N here (the ABI takes unsigned
// scalars; the reference's behaviour for them is whatever PBC's pow does, cmd/main.go:81).
func (e *Engine) EncryptBatch(x []*big.Int, r []*big.Int) ([]*Ciphertext, error) {
	xs := make([]*big.Int, len(x))
	for i, v := range x {
		xs[i] = new(big.Int).Mod(v, e.pk.N)
	}
	xb, xw := scalars(xs)
	rb, rw := e.randomness(r)
	out := make([]byte, len(x)*2*e.L)
	err := locked(func() C.int {
		return C.bgn_encrypt_batch(e.h, C.size_t(len(x)), u8(xb), C.size_t(xw), u8(rb), C.size_t(rw), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return e.unpack(out, false), nil
}

func (e *Engine) addSub(a, b []*Ciphertext, l2 bool, r []*big.Int, sub bool) ([]*Ciphertext, error) {
	A, B := e.pack(a), e.pack(b)
	rb, rw := e.randomness(r)
	out := make([]byte, len(A))
	err := locked(func() C.int {
		if sub {
			return C.bgn_sub_batch(e.h, C.size_t(len(a)), levelOf(l2), u8(A), u8(B), u8(rb), C.size_t(rw), u8(out))
		}
		return C.bgn_add_batch(e.h, C.size_t(len(a)), levelOf(l2), u8(A), u8(B), u8(rb), C.size_t(rw), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return e.unpack(out, l2), nil
}

// AddBatch is pk.Add (bgn.go:442-497) over arrays of one level; r: the blinding scalars of non-deterministic mode.
func (e *Engine) AddBatch(a, b []*Ciphertext, l2 bool, r []*big.Int) ([]*Ciphertext, error) {
	return e.addSub(a, b, l2, r, false)
}

// SubBatch is pk.Sub (bgn.go:375-433).
func (e *Engine) SubBatch(a, b []*Ciphertext, l2 bool, r []*big.Int) ([]*Ciphertext, error) {
	return e.addSub(a, b, l2, r, true)
}

// NegBatch is pk.Neg (bgn.go:436-438) in deterministic mode.
func (e *Engine) NegBatch(a []*Ciphertext, l2 bool) ([]*Ciphertext, error) {
	A := e.pack(a)
	out := make([]byte, len(A))
	if err := locked(func() C.int { return C.bgn_neg_batch(e.h, C.size_t(len(a)), levelOf(l2), u8(A), u8(out)) }); err != nil {
		return nil, err
	}
	return e.unpack(out, l2), nil
}

// MultBatch is pk.Mult (bgn.go:294-314) over arrays.  In non-deterministic mode the caller-supplied randomness
// r[i] < N replaces newCryptoRandom (bgn.go:303).
func (e *Engine) MultBatch(a, b []*Ciphertext, r []*big.Int) ([]*Ciphertext, error) {
	A, B := e.pack(a), e.pack(b)
	rb, rw := e.randomness(r)
	out := make([]byte, len(A))
	err := locked(func() C.int {
		return C.bgn_mult_batch(e.h, C.size_t(len(a)), u8(A), u8(B), u8(rb), C.size_t(rw), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return e.unpack(out, true), nil
}

// MakeL2Batch is pk.makeL2 (bgn.go:316-321).
func (e *Engine) MakeL2Batch(a []*Ciphertext) ([]*Ciphertext, error) {
	A := e.pack(a)
	out := make([]byte, len(A))
	if err := locked(func() C.int { return C.bgn_make_l2_batch(e.h, C.size_t(len(a)), u8(A), u8(out)) }); err != nil {
		return nil, err
	}
	return e.unpack(out, true), nil
}

// MultConstBatch is pk.MultConst (bgn.go:253-291) with one non-negative constant per ciphertext.
func (e *Engine) MultConstBatch(a []*Ciphertext, k []*big.Int, l2 bool, r []*big.Int) ([]*Ciphertext, error) {
	A := e.pack(a)
	kb, kw := scalars(k)
	rb, rw := e.randomness(r)
	out := make([]byte, len(A))
	err := locked(func() C.int {
		return C.bgn_multconst_batch(e.h, C.size_t(len(a)), levelOf(l2), u8(A), u8(kb), C.size_t(kw), u8(rb), C.size_t(rw), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return e.unpack(out, l2), nil
}

// DecryptBatch is sk.Decrypt (bgn.go:205-250): errs[i] is the reference's "cannot find discrete log; out of bounds"
// (gsbs.go:105) where the status says so.
func (e *Engine) DecryptBatch(cts []*Ciphertext, l2 bool) ([]*big.Int, []error, error) {
	A := e.pack(cts)
	m := make([]int64, len(cts))
	st := make([]uint8, len(cts))
	var mp *C.int64_t
	var sp *C.uint8_t
	if len(cts) > 0 {
		mp = (*C.int64_t)(unsafe.Pointer(&m[0]))
		sp = (*C.uint8_t)(unsafe.Pointer(&st[0]))
	}
	if err := locked(func() C.int { return C.bgn_decrypt_batch(e.h, C.size_t(len(cts)), levelOf(l2), u8(A), mp, sp) }); err != nil {
		return nil, nil, err
	}
	vals, errs := make([]*big.Int, len(cts)), make([]error, len(cts))
	for i := range cts {
		if st[i] != 0 {
			errs[i] = errNoDL
		} else {
			vals[i] = big.NewInt(m[i])
		}
	}
	return vals, errs, nil
}

// MultPolyBatch is pk.MultPoly (poly.go:123-156) over npoly pairs of level-1 polynomials of d1 and d2 coefficients:
// a and b hold the coefficients polynomial after polynomial; the result has d1+d2 coefficients per product (the last
// one the GT identity, as the reference's loop leaves it).  Deterministic form; a non-deterministic key blinds the
// result once with AddBatch(result, identities, l2 = true, r).
func (e *Engine) MultPolyBatch(npoly, d1, d2 int, a, b []*Ciphertext) ([]*Ciphertext, error) {
	A, B := e.pack(a), e.pack(b)
	out := make([]byte, npoly*(d1+d2)*2*e.L)
	err := locked(func() C.int {
		return C.bgn_poly_mult_batch(e.h, C.size_t(npoly), C.size_t(d1), C.size_t(d2), u8(A), u8(B), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return e.unpack(out, true), nil
}

// MultConstPolyOne is the double loop of MultConstPoly (poly.go:97-109) for one polynomial: the sign split,
// NewUnbalancedPlaintext and the final NegPoly stay as they are in Go (poly.go:71-96, :115-119).
func (e *Engine) MultConstPolyOne(ct *PolyCiphertext, poly *PolyPlaintext) (*PolyCiphertext, error) {
	A := e.pack(ct.Coefficients)
	k := make([]byte, poly.Degree*8) // digits of the encoded constant, big-endian, 8 bytes each
	for i := 0; i < poly.Degree; i++ {
		binary.BigEndian.PutUint64(k[8*i:], poly.Coefficients[i].Uint64())
	}
	out := make([]byte, (ct.Degree+poly.Degree)*2*e.L)
	err := locked(func() C.int {
		return C.bgn_poly_multconst_batch(e.h, 1, C.size_t(ct.Degree), C.size_t(poly.Degree), levelOf(ct.L2), u8(A), u8(k), 8, 0, u8(out))
	})
	if err != nil {
		return nil, err
	}
	return &PolyCiphertext{e.unpack(out, ct.L2), ct.Degree + poly.Degree, ct.ScaleFactor + poly.ScaleFactor, ct.L2}, nil
}

// EvalPolyBatch is pk.EvalPoly (poly.go:58-68) over polynomials of one degree and level.
func (e *Engine) EvalPolyBatch(cts []*PolyCiphertext) ([]*Ciphertext, error) {
	if len(cts) == 0 {
		return nil, nil
	}
	d, l2 := cts[0].Degree, cts[0].L2
	buf := make([]byte, 0, len(cts)*d*2*e.L)
	for _, ct := range cts {
		if ct.Degree != d || ct.L2 != l2 {
			return nil, errors.New("bgn_amd: EvalPolyBatch takes polynomials of one degree and level")
		}
		buf = append(buf, e.pack(ct.Coefficients[:d])...)
	}
	out := make([]byte, len(cts)*2*e.L)
	err := locked(func() C.int {
		return C.bgn_poly_eval_batch(e.h, C.size_t(len(cts)), C.size_t(d), levelOf(l2), u8(buf),
		C.uint64_t(e.pk.PolyEncodingParams.PolyBase), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return e.unpack(out, l2), nil
}

// CheckDecryptionProofBatch is pk.CheckDecryptionProof (gadgets.go:57-61) over arrays.
func (e *Engine) CheckDecryptionProofBatch(cts []*Ciphertext, proofs []*DecryptionProof) ([]bool, error) {
	A := e.pack(cts)
	vs, rs := make([]*big.Int, len(proofs)), make([]*big.Int, len(proofs))
	for i, pr := range proofs {
		vs[i], rs[i] = pr.Value, pr.Randomness
	}
	vb, vw := scalars(vs)
	rb, rw := scalars(rs)
	ok := make([]uint8, len(cts))
	err := locked(func() C.int {
		return C.bgn_check_decryption_proof_batch(e.h, C.size_t(len(cts)), u8(A), u8(vb), C.size_t(vw), u8(rb), C.size_t(rw), u8(ok))
	})
	if err != nil {
		return nil, err
	}
	return bools(ok), nil
}

// CheckPlaintextKnowledgeBatch is pk.CheckProofOfPlaintextKnoewledge (gadgets.go:65-77) over arrays: hash()
// (gadgets.go:80-96) stays in Go.
func (e *Engine) CheckPlaintextKnowledgeBatch(cts []*Ciphertext, proofs []*ProofOfPlaintextKnowledge) ([]bool, error) {
	A := e.pack(cts)
	N := make([]byte, 0, len(A))
	c := make([]byte, 0, 32*len(cts))
	dls := make([]*big.Int, len(proofs))
	for i, pr := range proofs {
		N = append(N, e.wire(pr.Nonce)...)
		c = append(c, leftPad(hash(pr).Bytes(), 32)...)
		dls[i] = pr.DL
	}
	dl, dw := scalars(dls)
	ok := make([]uint8, len(cts))
	err := locked(func() C.int {
		return C.bgn_check_plaintext_knowledge_batch(e.h, C.size_t(len(cts)), u8(A), u8(N), u8(c), 32, u8(dl), C.size_t(dw), u8(ok))
	})
	if err != nil {
		return nil, err
	}
	return bools(ok), nil
}

func bools(v []uint8) []bool {
	out := make([]bool, len(v))
	for i, x := range v {
		out[i] = x != 0
	}
	return out
}

// ---- arrays that stay on the device ---------------------------------------------------------------------------------------
//
// The batch methods above take and return Go values: every call copies its arrays over PCIe and converts them to and
// from pbc.Elements.  A chain like MultPoly -> AddPoly -> Decrypt (poly.go:123-207 -> bgn.go:205) only needs the
// plaintexts at its end: DeviceArray keeps the wire bytes of level-uniform ciphertexts on the GPU between calls.  The
// `_dev` entry points of the C ABI run on the null stream here (stream = nil), so calls on one engine execute in the
// order they are made and Download sees their results.  tests/cpp/cgo_shape.c runs the same chain from C.

// DeviceArray is `Count` elements of 2L wire bytes each (or any other fixed width: scalars, plaintexts) in the memory
// of the engine's GPU.  Free it when done; it is not garbage collected.
type DeviceArray struct {
	e     *Engine
	p     unsafe.Pointer
	Count int
	Width int // bytes per element
	L2    bool
}

// NewDeviceArray allocates count elements of width bytes on the engine's device (bgn_dev_alloc).
func (e *Engine) NewDeviceArray(count, width int, l2 bool) (*DeviceArray, error) {
	if count <= 0 || width <= 0 {
		return nil, errors.New("bgn_amd: empty device array")
	}
	var p unsafe.Pointer
	err := locked(func() C.int {
		p = C.bgn_dev_alloc(e.h, C.size_t(count*width))
		if p == nil {
			return C.BGN_E_NOMEM
		}
		return 0
	})
	if err != nil {
		return nil, err
	}
	return &DeviceArray{e: e, p: p, Count: count, Width: width, L2: l2}, nil
}

// Free gives the array back (bgn_dev_free).
func (d *DeviceArray) Free() {
	if d != nil && d.p != nil {
		C.bgn_dev_free(d.e.h, d.p)
		d.p = nil
	}
}

func (d *DeviceArray) u8() *C.uint8_t { return (*C.uint8_t)(d.p) }

// UploadBytes copies host bytes (Count * Width of them) into the array.
func (d *DeviceArray) UploadBytes(b []byte) error {
	if len(b) != d.Count*d.Width {
		return errors.New("bgn_amd: upload size does not match the device array")
	}
	if len(b) == 0 {
		return nil
	}
	return locked(func() C.int { return C.bgn_dev_upload(d.e.h, d.p, unsafe.Pointer(&b[0]), C.size_t(len(b))) })
}

// DownloadBytes copies the array to host memory, after every call issued on this engine before it.
func (d *DeviceArray) DownloadBytes() ([]byte, error) {
	b := make([]byte, d.Count*d.Width)
	if len(b) == 0 {
		return b, nil
	}
	err := locked(func() C.int { return C.bgn_dev_download(d.e.h, unsafe.Pointer(&b[0]), d.p, C.size_t(len(b))) })
	return b, err
}

// Upload packs ciphertexts of one level (Element.Bytes(), ciphertext.go:79) into a new device array.
func (e *Engine) Upload(cts []*Ciphertext, l2 bool) (*DeviceArray, error) {
	d, err := e.NewDeviceArray(len(cts), 2*e.L, l2)
	if err != nil {
		return nil, err
	}
	if err := d.UploadBytes(e.pack(cts)); err != nil {
		d.Free()
		return nil, err
	}
	return d, nil
}

// Download unpacks the array into ciphertexts (NewCiphertextFromBytes' element part, ciphertext.go:100).
func (d *DeviceArray) Download() ([]*Ciphertext, error) {
	b, err := d.DownloadBytes()
	if err != nil {
		return nil, err
	}
	return d.e.unpack(b, d.L2), nil
}

func (e *Engine) likeOut(n int, l2 bool) (*DeviceArray, error) { return e.NewDeviceArray(n, 2*e.L, l2) }

// uploadScalars puts big-endian scalars of one width into a device array (nil for a nil slice: deterministic mode).
func (e *Engine) uploadScalars(b []byte, count, width int) (*DeviceArray, error) {
	if b == nil {
		return nil, nil
	}
	d, err := e.NewDeviceArray(count, width, false)
	if err != nil {
		return nil, err
	}
	if err := d.UploadBytes(b); err != nil {
		d.Free()
		return nil, err
	}
	return d, nil
}

func (d *DeviceArray) ptrOrNil() *C.uint8_t {
	if d == nil {
		return nil
	}
	return d.u8()
}

// EncryptBatchDev is EncryptBatch with the ciphertexts left on the device (bgn.go:340-353; bgn_encrypt_batch_dev).
func (e *Engine) EncryptBatchDev(x []*big.Int, r []*big.Int) (*DeviceArray, error) {
	xs := make([]*big.Int, len(x))
	for i, v := range x {
		xs[i] = new(big.Int).Mod(v, e.pk.N)
	}
	xb, xw := scalars(xs)
	rb, rw := e.randomness(r)
	dx, err := e.uploadScalars(xb, len(x), xw)
	if err != nil {
		return nil, err
	}
	defer dx.Free()
	dr, err := e.uploadScalars(rb, len(x), rw)
	if err != nil {
		return nil, err
	}
	defer dr.Free()
	out, err := e.likeOut(len(x), false)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int {
		return C.bgn_encrypt_batch_dev(e.h, C.size_t(len(x)), dx.u8(), C.size_t(xw), dr.ptrOrNil(), C.size_t(rw), out.u8(), nil)
	})
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

func (e *Engine) addSubDev(a, b *DeviceArray, r []*big.Int, sub bool) (*DeviceArray, error) {
	if a.Count != b.Count || a.L2 != b.L2 {
		return nil, errors.New("bgn_amd: operand arrays differ in count or level")
	}
	rb, rw := e.randomness(r)
	dr, err := e.uploadScalars(rb, a.Count, rw)
	if err != nil {
		return nil, err
	}
	defer dr.Free()
	out, err := e.likeOut(a.Count, a.L2)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int {
		if sub {
			return C.bgn_sub_batch_dev(e.h, C.size_t(a.Count), levelOf(a.L2), a.u8(), b.u8(), dr.ptrOrNil(), C.size_t(rw), out.u8(), nil)
		}
		return C.bgn_add_batch_dev(e.h, C.size_t(a.Count), levelOf(a.L2), a.u8(), b.u8(), dr.ptrOrNil(), C.size_t(rw), out.u8(), nil)
	})
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

// AddBatchDev is pk.Add over device arrays (bgn.go:442-497; bgn_add_batch_dev).  Operands of one level: lift with
// MakeL2BatchDev first where the reference would (bgn.go:444-452).
func (e *Engine) AddBatchDev(a, b *DeviceArray, r []*big.Int) (*DeviceArray, error) {
	return e.addSubDev(a, b, r, false)
}

// SubBatchDev is pk.Sub over device arrays (bgn.go:375-432; bgn_sub_batch_dev).
func (e *Engine) SubBatchDev(a, b *DeviceArray, r []*big.Int) (*DeviceArray, error) {
	return e.addSubDev(a, b, r, true)
}

// NegBatchDev is pk.Neg over a device array (bgn.go:436-438; bgn_neg_batch_dev).
func (e *Engine) NegBatchDev(a *DeviceArray) (*DeviceArray, error) {
	out, err := e.likeOut(a.Count, a.L2)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int { return C.bgn_neg_batch_dev(e.h, C.size_t(a.Count), levelOf(a.L2), a.u8(), out.u8(), nil) })
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

// MultBatchDev is pk.Mult over device arrays of level-1 ciphertexts (bgn.go:294-314; bgn_mult_batch_dev).
func (e *Engine) MultBatchDev(a, b *DeviceArray, r []*big.Int) (*DeviceArray, error) {
	if a.Count != b.Count || a.L2 || b.L2 {
		return nil, errors.New("bgn_amd: Mult takes two level-1 arrays of one length")
	}
	rb, rw := e.randomness(r)
	dr, err := e.uploadScalars(rb, a.Count, rw)
	if err != nil {
		return nil, err
	}
	defer dr.Free()
	out, err := e.likeOut(a.Count, true)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int {
		return C.bgn_mult_batch_dev(e.h, C.size_t(a.Count), a.u8(), b.u8(), dr.ptrOrNil(), C.size_t(rw), out.u8(), nil)
	})
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

// MakeL2BatchDev is pk.makeL2 over a device array (bgn.go:316-321; bgn_make_l2_batch_dev).
func (e *Engine) MakeL2BatchDev(a *DeviceArray) (*DeviceArray, error) {
	out, err := e.likeOut(a.Count, true)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int { return C.bgn_make_l2_batch_dev(e.h, C.size_t(a.Count), a.u8(), out.u8(), nil) })
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

// MultConstBatchDev is pk.MultConst with one scalar per element (bgn.go:253-291; bgn_multconst_batch_dev).
func (e *Engine) MultConstBatchDev(a *DeviceArray, k []*big.Int, r []*big.Int) (*DeviceArray, error) {
	if len(k) != a.Count {
		return nil, errors.New("bgn_amd: one scalar per element")
	}
	ks := make([]*big.Int, len(k))
	for i, v := range k {
		ks[i] = new(big.Int).Mod(v, e.pk.N)
	}
	kb, kw := scalars(ks)
	rb, rw := e.randomness(r)
	dk, err := e.uploadScalars(kb, a.Count, kw)
	if err != nil {
		return nil, err
	}
	defer dk.Free()
	dr, err := e.uploadScalars(rb, a.Count, rw)
	if err != nil {
		return nil, err
	}
	defer dr.Free()
	out, err := e.likeOut(a.Count, a.L2)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int {
		return C.bgn_multconst_batch_dev(e.h, C.size_t(a.Count), levelOf(a.L2), a.u8(), dk.u8(), C.size_t(kw), dr.ptrOrNil(),
			C.size_t(rw), out.u8(), nil)
	})
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

// MultPolyBatchDev is MultPoly over npoly pairs of coefficient polynomials resident on the device: a holds npoly*d1
// level-1 coefficients, b npoly*d2; the result holds npoly*(d1+d2) level-2 coefficients, the last one of every
// polynomial the GT identity (poly.go:123-156; bgn_poly_mult_batch_dev).
func (e *Engine) MultPolyBatchDev(npoly, d1, d2 int, a, b *DeviceArray) (*DeviceArray, error) {
	if a.Count != npoly*d1 || b.Count != npoly*d2 || a.L2 || b.L2 {
		return nil, errors.New("bgn_amd: coefficient arrays do not match npoly*d1 / npoly*d2 level-1 elements")
	}
	out, err := e.likeOut(npoly*(d1+d2), true)
	if err != nil {
		return nil, err
	}
	err = locked(func() C.int {
		return C.bgn_poly_mult_batch_dev(e.h, C.size_t(npoly), C.size_t(d1), C.size_t(d2), a.u8(), b.u8(), out.u8(), nil)
	})
	if err != nil {
		out.Free()
		return nil, err
	}
	return out, nil
}

// DecryptBatchDev is sk.Decrypt over a device array: plaintexts and per-element errors as DecryptBatch returns them
// (bgn.go:205-250, gsbs.go:105; bgn_decrypt_batch_dev).  Only 9 bytes per ciphertext cross PCIe.
func (e *Engine) DecryptBatchDev(cts *DeviceArray) ([]*big.Int, []error, error) {
	dm, err := e.NewDeviceArray(cts.Count, 8, false)
	if err != nil {
		return nil, nil, err
	}
	defer dm.Free()
	ds, err := e.NewDeviceArray(cts.Count, 1, false)
	if err != nil {
		return nil, nil, err
	}
	defer ds.Free()
	err = locked(func() C.int {
		return C.bgn_decrypt_batch_dev(e.h, C.size_t(cts.Count), levelOf(cts.L2), cts.u8(), (*C.int64_t)(dm.p), ds.u8(), nil)
	})
	if err != nil {
		return nil, nil, err
	}
	mb, err := dm.DownloadBytes()
	if err != nil {
		return nil, nil, err
	}
	sb, err := ds.DownloadBytes()
	if err != nil {
		return nil, nil, err
	}
	vals := make([]*big.Int, cts.Count)
	errs := make([]error, cts.Count)
	for i := range vals {
		if sb[i] != 0 {
			errs[i] = errNoDL
			continue
		}
		vals[i] = big.NewInt(int64(binary.LittleEndian.Uint64(mb[8*i : 8*i+8]))) // the device writes native int64 (little-endian hosts)
	}
	return vals, errs, nil
}

// ValidateBatch says for every element whether it is a valid encoding for its level — components below p and on the
// curve (level 1) or of norm 1 (level 2).  The reference accepts any bytes (Element.SetBytes, ciphertext.go:100;
// PBC maps an invalid point to the identity without telling) and the batch operations do not validate: ciphertexts
// from an untrusted source go through this call first (bgn_validate_batch).
func (e *Engine) ValidateBatch(cts []*Ciphertext, l2 bool) ([]bool, error) {
	A := e.pack(cts)
	ok := make([]uint8, len(cts))
	err := locked(func() C.int { return C.bgn_validate_batch(e.h, C.size_t(len(cts)), levelOf(l2), u8(A), u8(ok)) })
	if err != nil {
		return nil, err
	}
	return bools(ok), nil
}

// ValidateBatchDev is ValidateBatch on a device array (bgn_validate_batch_dev).
func (e *Engine) ValidateBatchDev(a *DeviceArray) ([]bool, error) {
	dok, err := e.NewDeviceArray(a.Count, 1, false)
	if err != nil {
		return nil, err
	}
	defer dok.Free()
	err = locked(func() C.int {
		return C.bgn_validate_batch_dev(e.h, C.size_t(a.Count), levelOf(a.L2), a.u8(), dok.u8(), nil)
	})
	if err != nil {
		return nil, err
	}
	ok, err := dok.DownloadBytes()
	if err != nil {
		return nil, err
	}
	return bools(ok), nil
}

// Calibrate re-derives the batch-size crossovers between the pairing-kernel families from timed probes on this
// device (about a second at a 1024-bit key; call it once, after SetupDecryption, not concurrently with other calls):
// the eight crossovers in elements, -1 where not calibrated (bgn_ctx_calibrate).
func (e *Engine) Calibrate() ([8]int64, error) {
	var out [8]C.int64_t
	err := locked(func() C.int { return C.bgn_ctx_calibrate(e.h, &out[0]) })
	var res [8]int64
	for i := range res {
		res[i] = int64(out[i])
	}
	return res, err
}

// ---- the reference's single-element methods as count-1 calls --------------------------------------------------------------
// One cgo crossing per call, as with PBC.  The reference's methods do not return errors: a failing call panics with
// the library's message, as misuse does there (bgn.go:68,72,88,389; gsbs.go:57).

func (pk *PublicKey) blinding() []*big.Int {
	if pk.Deterministic {
		return nil
	}
	return []*big.Int{newCryptoRandom(pk.N)} // bgn.go:303, :263, :466, :488: drawn here, passed in
}

func must(cts []*Ciphertext, err error) *Ciphertext {
	if err != nil {
		panic(err.Error())
	}
	return cts[0]
}

// engineMult: the body of Mult (bgn.go:294-314).
func (pk *PublicKey) engineMult(ct1 *Ciphertext, ct2 *Ciphertext) *Ciphertext {
	if ct1.L2 || ct2.L2 {
		panic("both ciphertexts must be level 1") // bgn.go:296
	}
	return must(pk.engine().MultBatch([]*Ciphertext{ct1}, []*Ciphertext{ct2}, pk.blinding()))
}

// engineMakeL2: the body of makeL2 (bgn.go:316-321).
func (pk *PublicKey) engineMakeL2(ct *Ciphertext) *Ciphertext {
	return must(pk.engine().MakeL2Batch([]*Ciphertext{ct}))
}

// engineAdd: the body of Add (bgn.go:442-497), level lift included (:447-453).
func (pk *PublicKey) engineAdd(a *Ciphertext, b *Ciphertext) *Ciphertext {
	if a.L2 && !b.L2 {
		b = pk.engineMakeL2(b)
	}
	if !a.L2 && b.L2 {
		a = pk.engineMakeL2(a)
	}
	return must(pk.engine().AddBatch([]*Ciphertext{a}, []*Ciphertext{b}, a.L2, pk.blinding()))
}

// engineSub: the body of Sub (bgn.go:375-433).
func (pk *PublicKey) engineSub(a *Ciphertext, b *Ciphertext) *Ciphertext {
	if a.L2 && !b.L2 {
		b = pk.engineMakeL2(b)
	}
	if !a.L2 && b.L2 {
		a = pk.engineMakeL2(a)
	}
	return must(pk.engine().SubBatch([]*Ciphertext{a}, []*Ciphertext{b}, a.L2, pk.blinding()))
}

// engineNeg: the body of Neg (bgn.go:436-438): Sub(encryptZero(), c), whose level-1 zero is lifted for a level-2 c.
func (pk *PublicKey) engineNeg(c *Ciphertext) *Ciphertext {
	return pk.engineSub(pk.encryptZero(), c)
}

// engineMultConst: the body of MultConst (bgn.go:253-291); a negative constant is reduced modulo N.
func (pk *PublicKey) engineMultConst(c *Ciphertext, constant *big.Int) *Ciphertext {
	k := new(big.Int).Mod(constant, pk.N)
	return must(pk.engine().MultConstBatch([]*Ciphertext{c}, []*big.Int{k}, c.L2, pk.blinding()))
}

// engineEncryptWithRandomness: the body of EncryptWithRandomness (bgn.go:340-353); EncryptDeterministic (:325-331)
// passes r == nil.
func (pk *PublicKey) engineEncryptWithRandomness(x *big.Int, r *big.Int) *Ciphertext {
	var rs []*big.Int
	if r != nil {
		rs = []*big.Int{r}
	}
	return must(pk.engine().EncryptBatch([]*big.Int{x}, rs))
}

// engineDecrypt: the body of Decrypt (bgn.go:205-250), negative retry and the zero short-cut included.
func (sk *SecretKey) engineDecrypt(ct *Ciphertext, pk *PublicKey) (*big.Int, error) {
	vals, errs, err := pk.engine().DecryptBatch([]*Ciphertext{ct}, ct.L2)
	if err != nil {
		panic(err.Error()) // "DL tables not computed!" as gsbs.go:57
	}
	return vals[0], errs[0]
}

// ---- several GPUs from one process ------------------------------------------------------------------------------------------

// MultiEngine wraps one bgn_mctx: the same PublicKey on every listed GPU of the node.  A batch is cut into contiguous
// shards — by element, MultPoly by polynomial (poly.go:139-153: the accumulation stays on one device) — and every
// shard runs on its device from its own host thread; results land in the caller's arrays.
type MultiEngine struct {
	h  *C.bgn_mctx
	e0 *Engine // view of device 0's context: wire helpers
}

func (pk *PublicKey) NewMultiEngine(devices []int) (*MultiEngine, error) {
	p, l, err := parseA1(pk.PairingParams)
	if err != nil {
		return nil, err
	}
	if len(devices) == 0 {
		return nil, errors.New("bgn_amd: empty device list")
	}
	pb, nb := p.Bytes(), pk.N.Bytes()
	P, Q := pk.P.Bytes(), pk.Q.Bytes()
	det := 0
	if pk.Deterministic {
		det = 1
	}
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	var h *C.bgn_mctx
	err = locked(func() C.int {
		return C.bgn_mctx_create(&h, u8(pb), C.size_t(len(pb)), u8(nb), C.size_t(len(nb)), C.uint64_t(l), u8(P), u8(Q),
		C.int(det), (*C.int)(unsafe.Pointer(&devs[0])), C.int(len(devs)))
	})
	if err != nil {
		return nil, err
	}
	c0 := C.bgn_mctx_ctx(h, 0)
	return &MultiEngine{h: h, e0: &Engine{h: c0, L: int(C.bgn_fp_bytes(c0)), pk: pk}}, nil
}

func (m *MultiEngine) Close() { C.bgn_mctx_destroy(m.h) }

func (m *MultiEngine) SetupDecryption(sk *SecretKey) error {
	q1 := sk.Key.Bytes()
	if err := locked(func() C.int { return C.bgn_mctx_set_secret(m.h, u8(q1), C.size_t(len(q1))) }); err != nil {
		return err
	}
	return locked(func() C.int { return C.bgn_mctx_setup_decryption(m.h, C.uint64_t(m.e0.pk.MsgSpace.Uint64())) })
}

// MultBatch: pk.Mult over len(a) pairs, sharded over the GPUs (bgn.go:294-314).
func (m *MultiEngine) MultBatch(a, b []*Ciphertext, r []*big.Int) ([]*Ciphertext, error) {
	A, B := m.e0.pack(a), m.e0.pack(b)
	rb, rw := m.e0.randomness(r)
	out := make([]byte, len(A))
	err := locked(func() C.int {
		return C.bgn_mmult_batch(m.h, C.size_t(len(a)), u8(A), u8(B), u8(rb), C.size_t(rw), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return m.e0.unpack(out, true), nil
}

// DecryptBatch: sk.Decrypt over len(cts) ciphertexts, sharded over the GPUs (bgn.go:205-250).
func (m *MultiEngine) DecryptBatch(cts []*Ciphertext, l2 bool) ([]*big.Int, []error, error) {
	A := m.e0.pack(cts)
	vals64 := make([]int64, len(cts))
	st := make([]uint8, len(cts))
	var mp *C.int64_t
	var sp *C.uint8_t
	if len(cts) > 0 {
		mp = (*C.int64_t)(unsafe.Pointer(&vals64[0]))
		sp = (*C.uint8_t)(unsafe.Pointer(&st[0]))
	}
	if err := locked(func() C.int { return C.bgn_mdecrypt_batch(m.h, C.size_t(len(cts)), levelOf(l2), u8(A), mp, sp) }); err != nil {
		return nil, nil, err
	}
	vals, errs := make([]*big.Int, len(cts)), make([]error, len(cts))
	for i := range cts {
		if st[i] != 0 {
			errs[i] = errNoDL
		} else {
			vals[i] = big.NewInt(vals64[i])
		}
	}
	return vals, errs, nil
}

// MultPolyBatch: npoly independent pk.MultPoly products (poly.go:123-156); a polynomial never spans two GPUs.
func (m *MultiEngine) MultPolyBatch(npoly, d1, d2 int, a, b []*Ciphertext) ([]*Ciphertext, error) {
	A, B := m.e0.pack(a), m.e0.pack(b)
	out := make([]byte, npoly*(d1+d2)*2*m.e0.L)
	err := locked(func() C.int {
		return C.bgn_mpoly_mult_batch(m.h, C.size_t(npoly), C.size_t(d1), C.size_t(d2), u8(A), u8(B), u8(out))
	})
	if err != nil {
		return nil, err
	}
	return m.e0.unpack(out, true), nil
}

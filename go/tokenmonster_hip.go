//go:build hip

// tokenmonster_hip.go — the cgo binding a maintainer drops next to go/tokenmonster.go to run the tokenize path on an
// MI355X through libtokenmonster_hip.so (include/tokenmonster_hip.h).  Build with `-tags hip`; without the tag nothing
// changes.  The Go toolchain is not part of the image this library was developed in, so this file is compiled by nobody
// there: every call it makes is exercised through the same C ABI by examples/server_jobs.c (the job-1 / job-20 framing
// of training/tokenmonsterserver.go), examples/tokenize_file.c and the ctypes tests.
//
// Seams replaced (nothing else in the reference changes):
//   (*Vocab).Tokenize / Count / TokenizeToSerialized over many documents   go/tokenmonster.go:959, :971, :986
//   the goroutine fan-out of tokenmonsterserver jobs 1 and 20              training/tokenmonsterserver.go:363-378, :773-787
//   the scoring loop of the trainvocab worker                              training/trainvocab.go:925-1176
//   (*Vocab).Decode over many id streams, the streaming *Decoder, Save        go/tokenmonster.go:445, :552-700, :2602
// Every call borrows Go memory for its duration only (cgo pointer rule).  An error means: use the existing CPU path - EXCEPT
// ErrHipInput (TM_E_INPUT): the text / vocabulary pair is one the reference's own walk does not terminate on (a UTF-16 vocabulary
// with one-byte keys beside the delete token), so falling back to vocab.tokenize would hang; report it to the caller instead.
// Goroutines may call concurrently: each call takes a lane (stream + workspace) of the vocabulary and makes the
// vocabulary's device current on whatever OS thread the goroutine is on.  The only per-thread state of the library is the text of
// tm_last_error() and the 'current device' that tm_vocab_load / tm_dataset_upload use: the helper `locked` keeps a call and what
// depends on it on one OS thread.
package tokenmonster

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -ltokenmonster_hip
#include <stdlib.h>
#include "tokenmonster_hip.h"
#include "tm_build.h"
*/
import "C"

import (
	"errors"
	"os"
	"runtime"
	"unsafe"
)

// HipVocab is the device-resident twin of a *Vocab: the same .vocab bytes, flattened into the walk tables in HBM.
type HipVocab struct {
	h   *C.tm_vocab
	len int // vocab.Len(): decides 2- or 4-byte ids on the server wire (tokenmonsterserver.go:350-353)
	borrowed bool // the handle belongs to a HipVocabSet (Member): Close must not free it
}

// locked runs one library call and, if it failed, reads the library's THREAD-LOCAL error message on the same OS thread: between two
// cgo calls the scheduler may move a goroutine to another thread, where tm_last_error() would be empty or somebody else's.
// TM_E_NOSPACE is not an error for the callers below (they retry with the capacity reported).
func locked(f func() C.int) (C.int, error) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := f()
	if rc == C.TM_E_INPUT {
		return rc, ErrHipInput
	}
	if rc != C.TM_OK && rc != C.TM_E_NOSPACE {
		return rc, errors.New("tokenmonster_hip: " + C.GoString(C.tm_last_error()))
	}
	return rc, nil
}

// ErrHipInput: do NOT retry on the CPU path (see the file comment).
var ErrHipInput = errors.New("tokenmonster_hip: the walk does not advance on this text with this vocabulary (the reference loops forever on it)")

// HipDeviceCount reports the usable gfx950 devices (0: keep using the CPU path).
func HipDeviceCount() int { return int(C.tm_device_count()) }

// HipDeviceNumaNode: the NUMA node the device's PCIe root hangs under (-1: the host does not say).  The library's own pipeline workers
// run there for the length of a call; a host that fills pinned input buffers from its own goroutines pins those threads to the same node.
func HipDeviceNumaNode(device int) int { return int(C.tm_device_numa_node(C.int(device))) }

// Denormalize is Vocab.Denormalize (go/tokenmonster.go:445-462) for one byte string (a token, a decoded fragment): capcode decoding only,
// on the host (tm_denormalize); documents go through Decode.
func (hv *HipVocab) Denormalize(b []byte) ([]byte, error) {
	if len(b) == 0 {
		return []byte{}, nil
	}
	var out *C.uint8_t
	var n C.size_t
	if _, err := locked(func() C.int {
		return C.tm_denormalize((*C.uint8_t)(unsafe.Pointer(&b[0])), C.size_t(len(b)), C.tm_vocab_capcode(hv.h), &out, &n)
	}); err != nil {
		return nil, err
	}
	defer C.tm_free(unsafe.Pointer(out))
	res := make([]byte, int(n))
	copy(res, unsafe.Slice((*byte)(unsafe.Pointer(out)), int(n))) // (not C.GoBytes: its length is a C int, and n is a size_t)
	return res, nil
}

// LoadHip uploads the vocabulary file that Load (go/tokenmonster.go:2656) reads to `device`.
func LoadHip(filename string, device int) (*HipVocab, error) {
	b, err := os.ReadFile(filename)
	if err != nil {
		return nil, err
	}
	if len(b) == 0 {
		return nil, errors.New("tokenmonster_hip: empty vocabulary file")
	}
	// (tm_set_device + tm_vocab_load would be two cgo calls, and "current device" belongs to the OS thread: the goroutine may be on
	// another thread - another device - by the second one)
	var h *C.tm_vocab
	if _, err := locked(func() C.int {
		return C.tm_vocab_load_on((*C.uint8_t)(unsafe.Pointer(&b[0])), C.size_t(len(b)), C.int(device), &h)
	}); err != nil {
		return nil, err
	}
	return &HipVocab{h: h, len: int(C.tm_vocab_size(h))}, nil
}

func (hv *HipVocab) Close() {
	if hv.h != nil && !hv.borrowed {
		C.tm_vocab_free(hv.h)
	}
	hv.h = nil
}

// pack lays documents out the way the kernels read them: one contiguous buffer + offsets.
func pack(docs [][]byte) ([]byte, []uint64) {
	offsets := make([]uint64, len(docs)+1)
	total := 0
	for i, d := range docs {
		total += len(d)
		offsets[i+1] = uint64(total)
	}
	text := make([]byte, total+1)
	for i, d := range docs {
		copy(text[offsets[i]:], d)
	}
	return text, offsets
}

// TokenizeSerializedBatch is TokenizeToSerialized (go/tokenmonster.go:986) over many RAW documents in one call: the
// normalization pre-step (go :242-253), the walk and the packing to encodingLength bytes per id all run on the device,
// chunks of documents overlap their PCIe transfers with the kernels of other chunks.  encodingLength 0 picks 2 or 3 by
// len(reverse) like go :990-996.  Returns the serialized ids per document and the count of characters without a token.
func (hv *HipVocab) TokenizeSerializedBatch(docs [][]byte, encodingLength uint8) ([][]byte, []int, uint8, error) {
	n := len(docs)
	text, offsets := pack(docs)
	byteOff := make([]uint64, n+1)
	missing := make([]uint32, n+1)
	capBytes := uint64(len(text))*2 + 64
	var used C.uint32_t
	for {
		out := make([]byte, capBytes+1)
		rc, err := locked(func() C.int {
			return C.tm_tokenize_pipeline(hv.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])),
				C.uint32_t(n), 1, C.uint32_t(encodingLength), 0, 0, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(capBytes),
				(*C.uint64_t)(unsafe.Pointer(&byteOff[0])), (*C.uint32_t)(unsafe.Pointer(&missing[0])), &used, nil)
		})
		if err != nil {
			return nil, nil, 0, err
		}
		if rc == C.TM_E_NOSPACE { // the capacity required is reported in byteOff[n]
			capBytes = byteOff[n]
			continue
		}
		res := make([][]byte, n)
		miss := make([]int, n)
		for i := range docs {
			res[i] = out[byteOff[i]:byteOff[i+1]]
			miss[i] = int(missing[i])
		}
		return res, miss, uint8(used), nil
	}
}

// TokenizeBatch returns uint32 ids of ALREADY NORMALIZED documents (what vocab.tokenize, go :1017, takes).
func (hv *HipVocab) TokenizeBatch(normalized [][]byte) ([][]uint32, []int, error) {
	n := len(normalized)
	text, offsets := pack(normalized)
	capTok := uint64(len(text)/2 + 2*n + 64)
	tokOff := make([]uint64, n+1)
	missing := make([]uint32, n+1)
	for {
		out := make([]uint32, capTok+1)
		rc, err := locked(func() C.int {
			return C.tm_tokenize_batch(hv.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])),
				C.uint32_t(n), (*C.uint32_t)(unsafe.Pointer(&out[0])), C.uint64_t(capTok),
				(*C.uint64_t)(unsafe.Pointer(&tokOff[0])), (*C.uint32_t)(unsafe.Pointer(&missing[0])))
		})
		if err != nil {
			return nil, nil, err
		}
		if rc == C.TM_E_NOSPACE {
			capTok = tokOff[n]
			continue
		}
		res := make([][]uint32, n)
		miss := make([]int, n)
		for i := range normalized {
			res[i] = out[tokOff[i]:tokOff[i+1]]
			miss[i] = int(missing[i])
		}
		return res, miss, nil
	}
}

// CountBatch is Count (go :971) over many RAW documents: server job 20.
func (hv *HipVocab) CountBatch(docs [][]byte) ([]int, error) {
	n := len(docs)
	text, offsets := pack(docs)
	counts := make([]uint64, n+1)
	if _, err := locked(func() C.int {
		return C.tm_count_batch_raw(hv.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(n),
			(*C.uint64_t)(unsafe.Pointer(&counts[0])), nil)
	}); err != nil {
		return nil, err
	}
	res := make([]int, n)
	for i := range res {
		res[i] = int(counts[i])
	}
	return res, nil
}

// ServerEncodingLength is the rule of tokenmonsterserver job 1 (training/tokenmonsterserver.go:350-353): 2 bytes per
// id unless vocab.Len() > 65536, then 4 (not the 2/3 rule of TokenizeToSerialized's automatic mode).
func (hv *HipVocab) ServerEncodingLength() uint8 {
	if hv.len > 65536 {
		return 4
	}
	return 2
}

// In tokenmonsterserver.go job 1 (:355-383) the whole `if numBatches == 1 { ... } else { goroutines }` block becomes
//
//	bodies := splitBatches(data, numBatches)                    // the same length-prefixed parsing as :356-369
//	enc, _, _, err := hipVocab.TokenizeSerializedBatch(bodies, hipVocab.ServerEncodingLength())
//	for i := range results { results[i] = work{enc[i], err} }
//
// and job 20 (:767-787) `counts, err := hipVocab.CountBatch(bodies)`.  The response framing below them is unchanged.

// HipDataset is the normalized training dataset of trainvocab, resident in HBM for the whole run (trainvocab.go:1660-1665).
type HipDataset struct{ h *C.tm_dataset }

// UploadDataset copies the dataset to `device` (named in the call: the current device of an OS thread is not a property of a goroutine).
func UploadDataset(normalized []byte, device int) (*HipDataset, error) {
	if len(normalized) == 0 {
		return nil, errors.New("tokenmonster_hip: empty dataset")
	}
	var d *C.tm_dataset
	if _, err := locked(func() C.int {
		return C.tm_dataset_upload_on((*C.uint8_t)(unsafe.Pointer(&normalized[0])), C.uint64_t(len(normalized)), C.int(device), &d)
	}); err != nil {
		return nil, err
	}
	return &HipDataset{d}, nil
}
func (d *HipDataset) Close() { C.tm_dataset_free(d.h) }

// ScoreCandidate replaces the worker loop of training/trainvocab.go:925-1176 for one candidate vocabulary: tokens are the
// candidate's byte strings (single bytes included, as the worker's testVocab has them).  Tables are built with the rules
// of trainvocab.go:548-907 (tm_build_vocab), uploaded, and the strips are walked; stripOff == nil walks the whole dataset
// as one strip (:909-922).  scores[id] is scores[index].V of :1109-1162.
func ScoreCandidate(d *HipDataset, tokens [][]byte, capcode, charset uint8, stripOff, stripLen []uint64) (scores []uint32, tokensInText uint64, missing [32]byte, err error) {
	blob, off64 := pack(tokens)
	off := make([]uint32, len(off64))
	for i, o := range off64 {
		off[i] = uint32(o)
	}
	// token list -> records -> walk tables -> the dataset's device in one call (tm_vocab_build; rounds 2-3 wrote a .vocab image with
	// tm_build_vocab and parsed it again in tm_vocab_load: twice the host time)
	var cand *C.tm_vocab
	if _, err = locked(func() C.int {
		return C.tm_vocab_build((*C.uint8_t)(unsafe.Pointer(&blob[0])), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(tokens)), nil,
			C.uint32_t(capcode), C.uint32_t(charset), 0, 5, 0, C.tm_dataset_device(d.h), &cand)
	}); err != nil {
		return nil, 0, missing, err
	}
	defer C.tm_vocab_free(cand)
	scores = make([]uint32, int(C.tm_vocab_n_ids(cand))+1)
	var so, sl *C.uint64_t
	if len(stripOff) > 0 {
		so = (*C.uint64_t)(unsafe.Pointer(&stripOff[0]))
		sl = (*C.uint64_t)(unsafe.Pointer(&stripLen[0]))
	}
	var tit C.uint64_t
	if _, err = locked(func() C.int {
		return C.tm_score(cand, d.h, so, sl, C.uint32_t(len(stripOff)), (*C.uint32_t)(unsafe.Pointer(&scores[0])), &tit,
			(*C.uint8_t)(unsafe.Pointer(&missing[0])))
	}); err != nil {
		return nil, 0, missing, err
	}
	return scores[:len(scores)-1], uint64(tit), missing, nil
}

// ---- the Detokenize half: Decode (go/tokenmonster.go:445) and the streaming Decoder (:552-700) --------------------------------

// DecodeBatch is (*Vocab).Decode over many id streams in one call: the gather of the token bytes runs on the device, capcode
// decoding follows (raw == true skips it, like the C++ runtime's decode_raw).
func (hv *HipVocab) DecodeBatch(tokens [][]uint32, raw bool) ([][]byte, error) {
	n := len(tokens)
	tokOff := make([]uint64, n+1)
	total := 0
	for i, t := range tokens {
		total += len(t)
		tokOff[i+1] = uint64(total)
	}
	flat := make([]uint32, total+1)
	for i, t := range tokens {
		copy(flat[tokOff[i]:], t)
	}
	outOff := make([]uint64, n+1)
	capBytes := uint64(total)*6 + 64
	r := C.int(0)
	if raw {
		r = 1
	}
	for {
		out := make([]byte, capBytes+1)
		rc, err := locked(func() C.int {
			return C.tm_decode_batch(hv.h, (*C.uint32_t)(unsafe.Pointer(&flat[0])), (*C.uint64_t)(unsafe.Pointer(&tokOff[0])), C.uint32_t(n), r,
				(*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(capBytes), (*C.uint64_t)(unsafe.Pointer(&outOff[0])))
		})
		if err != nil {
			return nil, err
		}
		if rc == C.TM_E_NOSPACE { // the capacity required is reported in outOff[n]
			capBytes = outOff[n]
			continue
		}
		res := make([][]byte, n)
		for i := range res {
			res[i] = out[outOff[i]:outOff[i+1]]
		}
		return res, nil
	}
}

// HipDecoder is the twin of *Decoder (go/tokenmonster.go:552 NewDecoder): ids arrive a few at a time, Decode returns the text that
// is complete so far and keeps the bytes of a cut UTF-8 character and the capcode state for the next call; Flush hands back the rest.
type HipDecoder struct{ h *C.tm_decoder }

func (hv *HipVocab) NewDecoder() (*HipDecoder, error) {
	var d *C.tm_decoder
	if _, err := locked(func() C.int { return C.tm_decoder_new(hv.h, &d) }); err != nil {
		return nil, err
	}
	return &HipDecoder{d}, nil
}
func (d *HipDecoder) Close() { C.tm_decoder_free(d.h) }

func (d *HipDecoder) Decode(tokens []uint32) ([]byte, error) {
	var tp *C.uint32_t
	if len(tokens) > 0 {
		tp = (*C.uint32_t)(unsafe.Pointer(&tokens[0]))
	}
	out := make([]byte, len(tokens)*48+64)
	var n C.uint64_t
	rc, err := locked(func() C.int {
		return C.tm_decoder_decode(d.h, tp, C.uint64_t(len(tokens)), (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(len(out)), &n)
	})
	if err == nil && rc == C.TM_E_NOSPACE { // the ids HAVE been consumed and the text is kept: fetch it with n = 0 and a buffer of the size reported
		out = make([]byte, uint64(n)+1)
		_, err = locked(func() C.int {
			return C.tm_decoder_decode(d.h, nil, 0, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(len(out)), &n)
		})
	}
	if err != nil {
		return nil, err
	}
	return out[:n], nil
}

func (d *HipDecoder) Flush() ([]byte, error) {
	out := make([]byte, 64)
	var n C.uint64_t
	rc, err := locked(func() C.int {
		return C.tm_decoder_flush(d.h, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(len(out)), &n)
	})
	if err == nil && rc == C.TM_E_NOSPACE { // the remainder is held back (it has been moved to the pending text): fetch everything with a buffer of the size reported
		out = make([]byte, uint64(n)+1)
		_, err = locked(func() C.int {
			return C.tm_decoder_decode(d.h, nil, 0, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(len(out)), &n)
		})
	}
	if err != nil {
		return nil, err
	}
	return out[:n], nil
}

// Save writes the vocabulary file back (go/tokenmonster.go:2602 Save): this library never mutates a vocabulary, so the bytes are the
// ones LoadHip read.
func (hv *HipVocab) Save(filename string) error {
	cs := C.CString(filename)
	defer C.free(unsafe.Pointer(cs))
	_, err := locked(func() C.int { return C.tm_vocab_save(hv.h, cs) })
	return err
}

// Tune lays the vocabulary's device tables out by use on a sample of NORMALIZED text (tm_vocab_tune): worth it for large vocabularies
// (100 000 ids: -5 % on the match kernel), changes no result, and must not run beside other calls on this vocabulary.
func (hv *HipVocab) Tune(normalizedSample []byte) error {
	var p *C.uint8_t
	if len(normalizedSample) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&normalizedSample[0]))
	}
	_, err := locked(func() C.int { return C.tm_vocab_tune(hv.h, p, C.uint64_t(len(normalizedSample))) })
	return err
}

// ---- trainvocab over several GPUs: one process (or one locked OS thread) per device, each owning a byte range of the dataset -------
// ScoreRangeBegin / ScoreRangeFinish are the two halves of a pass over the range [0, ownLen) of a dataset that was uploaded followed by
// >= 128 bytes of the text that comes next (continues == true): Begin returns what the range does to each of the 80 entry states; the
// ranks exchange those 80 bytes, rank r chains the maps of the ranks before it from state 0 to its own entry state, and Finish
// completes the pass from there.  Summed over the ranks the histograms equal ONE walk over the whole dataset (trainvocab.go:909-922).
// textEndsInHalo: text follows the range, but fewer than 128 bytes of it - and the dataset holds all of them (continues = 2 in the C ABI).
func ScoreRangeBegin(cand *HipVocab, d *HipDataset, ownLen uint64, continues bool, textEndsInHalo bool) (exits [80]byte, err error) {
	c := C.int(0)
	if continues {
		c = 1
		if textEndsInHalo {
			c = 2
		}
	}
	_, err = locked(func() C.int {
		return C.tm_score_begin(cand.h, d.h, 0, C.uint64_t(ownLen), c, nil, (*C.uint8_t)(unsafe.Pointer(&exits[0])))
	})
	return exits, err
}

func ScoreRangeFinish(cand *HipVocab, d *HipDataset, entryState uint32) (scores []uint32, tokensInText uint64, missing [32]byte, err error) {
	if _, err = locked(func() C.int { return C.tm_score_finish(cand.h, d.h, C.uint32_t(entryState), nil, nil, 0) }); err != nil {
		return nil, 0, missing, err
	}
	scores = make([]uint32, int(C.tm_vocab_n_ids(cand.h))+1)
	var tit C.uint64_t
	if _, err = locked(func() C.int {
		return C.tm_score_read(cand.h, d.h, (*C.uint32_t)(unsafe.Pointer(&scores[0])), &tit, (*C.uint8_t)(unsafe.Pointer(&missing[0])))
	}); err != nil {
		return nil, 0, missing, err
	}
	return scores[:len(scores)-1], uint64(tit), missing, nil
}

// ---- every GPU of the node from ONE process (include/tokenmonster_hip.h, "several devices") ---------------------------------------
// The reference's own parallelism is in-process: tokenmonsterserver fans a job's documents out over goroutines
// (training/tokenmonsterserver.go:363-378), trainvocab starts `workers` goroutines over one dataset (training/trainvocab.go:1827-1829).
// HipDevices is the handle of the node's GPUs; the library drives them itself (one host thread per device) and does the one collective of
// the path - the all-reduce of the scoring pass's histogram - with RCCL inside tm_score_multi.  Nothing here needs LockOSThread beyond
// locked(): the library makes the right device current on whatever thread it runs.
type HipDevices struct{ h *C.tm_devices }
type HipVocabSet struct{ h *C.tm_vocab_set }
type HipDatasetSet struct{ h *C.tm_dataset_set }

// OpenHipDevices(0) = every visible GPU.
func OpenHipDevices(maxDevices int) (*HipDevices, error) {
	var h *C.tm_devices
	if _, err := locked(func() C.int { return C.tm_devices_open(C.int(maxDevices), &h) }); err != nil {
		return nil, err
	}
	return &HipDevices{h}, nil
}
func (g *HipDevices) Count() int { return int(C.tm_devices_count(g.h)) }
func (g *HipDevices) Close()     { C.tm_devices_close(g.h) }

// LoadHipAll is LoadHip for every device of the handle: the tables are built once, the finished device block is replicated over xGMI.
func LoadHipAll(g *HipDevices, filename string) (*HipVocabSet, error) {
	b, err := os.ReadFile(filename)
	if err != nil {
		return nil, err
	}
	return NewHipVocabSet(g, b)
}

// BuildHipVocabImage turns a candidate's token list into the bytes of a .vocab file with the rules of trainvocab.go:548-907
// (tm_build_vocab, include/tm_build.h): what ScoreCandidate does before it loads the tables.
func BuildHipVocabImage(tokens [][]byte, capcode, charset uint8) ([]byte, error) {
	blob, off64 := pack(tokens)
	off := make([]uint32, len(off64))
	for i, o := range off64 {
		off[i] = uint32(o)
	}
	var img *C.uint8_t
	var imgLen C.size_t
	if _, err := locked(func() C.int {
		return C.tm_build_vocab((*C.uint8_t)(unsafe.Pointer(&blob[0])), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(tokens)), nil,
			C.uint32_t(capcode), C.uint32_t(charset), 0, 5, 0, &img, &imgLen)
	}); err != nil {
		return nil, err
	}
	defer C.tm_free(unsafe.Pointer(img))
	return C.GoBytes(unsafe.Pointer(img), C.int(imgLen)), nil
}

// NewHipVocabSet: the same from the bytes of a .vocab image (a candidate from BuildHipVocabImage in the trainvocab worker).
func NewHipVocabSet(g *HipDevices, image []byte) (*HipVocabSet, error) {
	if len(image) == 0 {
		return nil, errors.New("tokenmonster_hip: empty vocabulary image")
	}
	var h *C.tm_vocab_set
	if _, err := locked(func() C.int {
		return C.tm_vocab_load_all(g.h, (*C.uint8_t)(unsafe.Pointer(&image[0])), C.size_t(len(image)), &h)
	}); err != nil {
		return nil, err
	}
	return &HipVocabSet{h}, nil
}
func (s *HipVocabSet) Close() { C.tm_vocab_set_free(s.h) }

// Member(0) is a full vocabulary (Decode, NewDecoder, Save); the handle stays owned by the set: Close on the returned value does
// nothing (tm_vocab_set_free frees the members), and it must not be used after the set has been closed.  nil: no such member.
func (s *HipVocabSet) Member(i int) *HipVocab {
	h := (*C.tm_vocab)(unsafe.Pointer(C.tm_vocab_set_member(s.h, C.int(i))))
	if h == nil {
		return nil
	}
	return &HipVocab{h: h, len: int(C.tm_vocab_size(h)), borrowed: true}
}

// TokenizeSerializedBatch over every GPU: same arguments and results as (*HipVocab).TokenizeSerializedBatch - server job 1 does not change.
func (s *HipVocabSet) TokenizeSerializedBatch(docs [][]byte, encodingLength uint8) ([][]byte, []int, uint8, error) {
	text, offsets := pack(docs)
	n := len(docs)
	capBytes := uint64(len(text))*2 + uint64(8*n) + 64
	byteOff := make([]uint64, n+1)
	missing := make([]uint32, n+1)
	var used C.uint32_t
	for {
		out := make([]byte, capBytes+1)
		rc, err := locked(func() C.int {
			return C.tm_tokenize_pipeline_multi(s.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(n), 1,
				C.uint32_t(encodingLength), 0, 0, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(capBytes), (*C.uint64_t)(unsafe.Pointer(&byteOff[0])),
				(*C.uint32_t)(unsafe.Pointer(&missing[0])), &used, nil)
		})
		if err != nil {
			return nil, nil, 0, err
		}
		if rc == C.TM_E_NOSPACE {
			capBytes = byteOff[n]
			continue
		}
		res := make([][]byte, n)
		miss := make([]int, n)
		for i := 0; i < n; i++ {
			res[i] = out[byteOff[i]:byteOff[i+1]]
			miss[i] = int(missing[i])
		}
		return res, miss, uint8(used), nil
	}
}

// UploadHipDatasetSharded: once per training run, after `filedata = normalize(ReadFile(dataset))` (trainvocab.go:1660-1665): one byte
// range (+ 128 bytes of halo) per GPU.
func UploadHipDatasetSharded(g *HipDevices, normalized []byte) (*HipDatasetSet, error) {
	var h *C.tm_dataset_set
	var p *C.uint8_t
	if len(normalized) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&normalized[0]))
	}
	if _, err := locked(func() C.int { return C.tm_dataset_upload_sharded(g.h, p, C.uint64_t(len(normalized)), &h) }); err != nil {
		return nil, err
	}
	return &HipDatasetSet{h}, nil
}
func (d *HipDatasetSet) Close() { C.tm_dataset_set_free(d.h) }

// ScoreCandidateAll replaces the worker's inner loop (trainvocab.go:925-1176) for the post-"midway" mode (:909-922: the dataset as ONE
// strip) on every GPU of the node: scores[id] == scores[index].V of :1109-1162 summed over the whole dataset, bit-identical to ScoreCandidate
// on one GPU.  The histogram all-reduce is RCCL inside the call.
func ScoreCandidateAll(cand *HipVocabSet, d *HipDatasetSet) (scores []uint32, tokensInText uint64, missing [32]byte, err error) {
	nIds := int(C.tm_vocab_n_ids(C.tm_vocab_set_member(cand.h, 0)))
	scores = make([]uint32, nIds+1)
	var tit C.uint64_t
	if _, err = locked(func() C.int {
		return C.tm_score_multi(cand.h, d.h, (*C.uint32_t)(unsafe.Pointer(&scores[0])), &tit, (*C.uint8_t)(unsafe.Pointer(&missing[0])))
	}); err != nil {
		return nil, 0, missing, err
	}
	return scores[:nIds], uint64(tit), missing, nil
}

// NewHipVocabSetFromTokens: a candidate's tables for every GPU straight from its token list (tm_vocab_build_all): built once on the host,
// uploaded to the first device, replicated device to device.
func NewHipVocabSetFromTokens(g *HipDevices, tokens [][]byte, capcode, charset uint8) (*HipVocabSet, error) {
	blob, off64 := pack(tokens)
	off := make([]uint32, len(off64))
	for i, o := range off64 {
		off[i] = uint32(o)
	}
	var h *C.tm_vocab_set
	if _, err := locked(func() C.int {
		return C.tm_vocab_build_all(g.h, (*C.uint8_t)(unsafe.Pointer(&blob[0])), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(tokens)), nil,
			C.uint32_t(capcode), C.uint32_t(charset), 0, 5, 0, &h)
	}); err != nil {
		return nil, err
	}
	return &HipVocabSet{h}, nil
}

//go:build hip

// tokenmonster_hip_resident.go — the rest of the C ABI of libtokenmonster_hip.so (include/tokenmonster_hip.h, include/tm_build.h) for a Go
// host: what tokenmonster_hip.go does not need for its drop-in seams but a service built around the library does — page-locked buffers for
// the host-to-host ring, the device-resident batch (text and ids stay in HBM between normalize, tokenize and decode), vocabulary accessors,
// the host normalizer, .tok dictionaries, device blocks, the asynchronous scoring entry points.  Like its sibling this file has never seen a
// compiler in the image the library was developed in (no Go toolchain there); tests/test_abi.py checks that every tm_* symbol the headers
// declare is bound in one of the two files, so that an ABI change cannot leave the Go side behind unnoticed.
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
	"unsafe"
)

// ---- vocabulary accessors (go/tokenmonster.go:2390-2500) --------------------------------------------------------------------------------------

func (hv *HipVocab) Charset() uint8           { return uint8(C.tm_vocab_charset(hv.h)) }
func (hv *HipVocab) NormalizationCode() uint8 { return uint8(C.tm_vocab_normalization(hv.h)) }
func (hv *HipVocab) MaxTokenLength() int      { return int(C.tm_vocab_max_token_length(hv.h)) }
func (hv *HipVocab) NumRecords() int          { return int(C.tm_vocab_n_info(hv.h)) }
func (hv *HipVocab) DeviceBytes() uint64      { return uint64(C.tm_vocab_device_bytes(hv.h)) }

// Unk / DeleteToken: the id, and whether the vocabulary has one (TM_NONE otherwise).
func (hv *HipVocab) Unk() (uint32, bool) {
	u := uint32(C.tm_vocab_unk(hv.h))
	return u, u != uint32(C.TM_NONE)
}
func (hv *HipVocab) DeleteToken() (uint32, bool) {
	u := uint32(C.tm_vocab_delete_token(hv.h))
	return u, u != uint32(C.TM_NONE)
}

// Image returns the .vocab bytes of the vocabulary (what Save writes, go :2602): a copy, the library keeps its own.
func (hv *HipVocab) Image() ([]byte, error) {
	var p *C.uint8_t
	var n C.size_t
	if _, err := locked(func() C.int { return C.tm_vocab_image(hv.h, &p, &n) }); err != nil {
		return nil, err
	}
	return C.GoBytes(unsafe.Pointer(p), C.int(n)), nil
}

// HipSetDevice makes `device` current for the calling OS thread (the one-device entry points that take no vocabulary use it); call it
// inside runtime.LockOSThread.  HipKernelName names the stages RunTimed reports.
func HipSetDevice(device int) error {
	_, err := locked(func() C.int { return C.tm_set_device(C.int(device)) })
	return err
}
func HipKernelName(k int) string { return C.GoString(C.tm_kernel_name(C.int(k))) }

// HipDecodeHostDocs: documents of the calling OS thread's last DecodeBatch that the device left to the host decoder (call it on the same
// locked thread as the DecodeBatch it asks about).
func HipDecodeHostDocs() int { return int(C.tm_decode_host_docs()) }

// hipTestHooks arms nothing in a process that was not started with TM_TEST_HOOKS in its environment; it is here for the binding's own tests.
func hipTestHooks(flags int) int { return int(C.tm_debug_flags(C.int(flags))) }

// ---- page-locked host memory: what the host-to-host ring wants on both sides (tm_tokenize_pipeline) -----------------------------------------

// HipHostBuffer is page-locked memory from the HIP runtime, placed on the NUMA node nearest to the current device.  Bytes() is valid until Free.
type HipHostBuffer struct {
	p unsafe.Pointer
	n int
}

func AllocHipHostBuffer(bytes int) (*HipHostBuffer, error) {
	p := C.tm_host_alloc(C.size_t(bytes))
	if p == nil {
		return nil, errors.New("tokenmonster_hip: page-locked allocation failed")
	}
	return &HipHostBuffer{p, bytes}, nil
}
func (b *HipHostBuffer) Bytes() []byte { return unsafe.Slice((*byte)(b.p), b.n) }
func (b *HipHostBuffer) Free()         { C.tm_host_free(b.p); b.p = nil }

// RegisterHipHostMemory page-locks memory the caller owns (it must not move: C memory, or a Go slice pinned for as long as it is registered).
func RegisterHipHostMemory(p unsafe.Pointer, bytes int) error {
	_, err := locked(func() C.int { return C.tm_host_register(p, C.size_t(bytes)) })
	return err
}
func UnregisterHipHostMemory(p unsafe.Pointer) error {
	_, err := locked(func() C.int { return C.tm_host_unregister(p) })
	return err
}

// TokenizeSerializedPinned is TokenizeSerializedBatch on page-locked buffers the caller keeps between calls: raw documents packed in `text`
// (offsets[len] = bytes), serialized ids into `out`; both DMA'd directly - the ring of tm_host.hip, no host round trip inside a chunk.
// Returns the byte offsets per document, the missing counts and the id width used; ErrHipNoSpace with the size required if `out` is too small.
var ErrHipNoSpace = errors.New("tokenmonster_hip: output buffer too small")

func (hv *HipVocab) TokenizeSerializedPinned(text *HipHostBuffer, offsets []uint64, encodingLength uint8, out *HipHostBuffer) ([]uint64, []uint32, uint8, uint64, error) {
	n := len(offsets) - 1
	byteOff := make([]uint64, n+1)
	missing := make([]uint32, n+1)
	var used C.uint32_t
	var stats C.tm_pipeline_stats
	rc, err := locked(func() C.int {
		return C.tm_tokenize_pipeline(hv.h, (*C.uint8_t)(text.p), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(n), 1, C.uint32_t(encodingLength), 0, 0,
			(*C.uint8_t)(out.p), C.uint64_t(out.n), (*C.uint64_t)(unsafe.Pointer(&byteOff[0])), (*C.uint32_t)(unsafe.Pointer(&missing[0])), &used, &stats)
	})
	if err != nil {
		return nil, nil, 0, 0, err
	}
	if rc == C.TM_E_NOSPACE {
		return nil, nil, 0, byteOff[n], ErrHipNoSpace
	}
	return byteOff, missing[:n], uint8(used), byteOff[n], nil
}

// TokenizeNormalizedSerialized is TokenizeToSerialized on ALREADY NORMALIZED documents in one resident batch (tm_tokenize_batch_serialized).
func (hv *HipVocab) TokenizeNormalizedSerialized(normalized [][]byte, encodingLength uint8) ([][]byte, []int, uint8, error) {
	n := len(normalized)
	text, offsets := pack(normalized)
	byteOff := make([]uint64, n+1)
	missing := make([]uint32, n+1)
	capBytes := uint64(len(text))*2 + 64
	var used C.uint32_t
	for {
		out := make([]byte, capBytes+1)
		rc, err := locked(func() C.int {
			return C.tm_tokenize_batch_serialized(hv.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(n),
				C.uint32_t(encodingLength), (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(capBytes), (*C.uint64_t)(unsafe.Pointer(&byteOff[0])),
				(*C.uint32_t)(unsafe.Pointer(&missing[0])), &used)
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
		for i := range res {
			res[i] = out[byteOff[i]:byteOff[i+1]]
			miss[i] = int(missing[i])
		}
		return res, miss, uint8(used), nil
	}
}

// ---- the device-resident batch: text and ids stay in HBM between the stages (what bench.py times) ------------------------------------------

// HipBatch owns device buffers for up to maxBytes of text in up to maxDocs documents.  Not for concurrent use; one batch per goroutine.
type HipBatch struct{ h *C.tm_batch }

func (hv *HipVocab) NewBatch(maxBytes uint64, maxDocs int) (*HipBatch, error) {
	var b *C.tm_batch
	if _, err := locked(func() C.int { return C.tm_batch_create(hv.h, C.uint64_t(maxBytes), C.uint32_t(maxDocs), &b) }); err != nil {
		return nil, err
	}
	return &HipBatch{b}, nil
}
func (b *HipBatch) Close() { C.tm_batch_free(b.h); b.h = nil }

// Upload: normalized documents; UploadRaw: raw UTF-8, to be normalized on the device by Normalize.
func (b *HipBatch) Upload(text []byte, offsets []uint64) error {
	_, err := locked(func() C.int {
		return C.tm_batch_upload(b.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(len(offsets)-1))
	})
	return err
}
func (b *HipBatch) UploadRaw(raw []byte, offsets []uint64) error {
	_, err := locked(func() C.int {
		return C.tm_batch_upload_raw(b.h, (*C.uint8_t)(unsafe.Pointer(&raw[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(len(offsets)-1))
	})
	return err
}

// Normalize runs norm.Normalize + capcode.Encode on the device (every normalization flag; documents the device cannot do go through the host
// normalizer inside the call: HostFallbackDocs says how many); Run the tokenizer pipeline.  Both on the NULL stream of the batch's device.
func (b *HipBatch) Normalize() error {
	_, err := locked(func() C.int { return C.tm_batch_normalize(b.h, nil) })
	return err
}
func (b *HipBatch) NormalizedBytes() uint64 { return uint64(C.tm_batch_normalized_bytes(b.h)) }
func (b *HipBatch) HostFallbackDocs() int   { return int(C.tm_batch_host_fallback_docs(b.h)) }
func (b *HipBatch) DeviceBytes() uint64     { return uint64(C.tm_batch_device_bytes(b.h)) }
func (b *HipBatch) Run() error {
	_, err := locked(func() C.int { return C.tm_batch_run(b.h, nil) })
	return err
}

// RunTimed: milliseconds per stage (HipKernelName(0..4)), by HIP events on the launch stream.
func (b *HipBatch) RunTimed() ([5]float32, error) {
	var ms [5]C.float
	_, err := locked(func() C.int { return C.tm_batch_run_timed(b.h, nil, &ms[0]) })
	var out [5]float32
	for i := range out {
		out[i] = float32(ms[i])
	}
	return out, err
}
func (b *HipBatch) Totals() (tokens, missing uint64, err error) {
	var t, m C.uint64_t
	_, err = locked(func() C.int { return C.tm_batch_totals(b.h, &t, &m) })
	return uint64(t), uint64(m), err
}

// Download: the ids of the last Run, their offsets per document and the missing counts.  DeviceTokens / DeviceTokOffsets: where they lie in
// HBM (valid until the next Run; call Totals first - it grows the id buffer and repeats the emit stage if the run outgrew it).
func (b *HipBatch) Download(ndocs int) ([]uint32, []uint64, []uint32, error) {
	tokens, _, err := b.Totals()
	if err != nil {
		return nil, nil, nil, err
	}
	ids := make([]uint32, tokens+1)
	off := make([]uint64, ndocs+1)
	miss := make([]uint32, ndocs+1)
	_, err = locked(func() C.int {
		return C.tm_batch_download(b.h, (*C.uint32_t)(unsafe.Pointer(&ids[0])), C.uint64_t(tokens), (*C.uint64_t)(unsafe.Pointer(&off[0])), (*C.uint32_t)(unsafe.Pointer(&miss[0])))
	})
	return ids[:tokens], off, miss[:ndocs], err
}
func (b *HipBatch) DeviceTokens() unsafe.Pointer     { return unsafe.Pointer(C.tm_batch_device_tokens(b.h)) }
func (b *HipBatch) DeviceTokOffsets() unsafe.Pointer { return unsafe.Pointer(C.tm_batch_device_tok_offsets(b.h)) }

// DownloadText: the normalized text of the batch in document order (a check, not a hot path).
func (b *HipBatch) DownloadText(ndocs int) ([]byte, []uint64, error) {
	n := b.NormalizedBytes()
	text := make([]byte, n+1)
	off := make([]uint64, ndocs+1)
	_, err := locked(func() C.int {
		return C.tm_batch_download_text(b.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), C.uint64_t(n), (*C.uint64_t)(unsafe.Pointer(&off[0])))
	})
	return text[:n], off, err
}

// Decode decodes the ids the batch holds where they lie (ids in HBM -> text in HBM): bytes the device decoded and documents it left to the
// host decoder; DecodeTimed adds the milliseconds of its three stages; DecodedText fetches the text of every document (the host decoder's included).
func (b *HipBatch) Decode(raw bool) (uint64, int, error) {
	var n C.uint64_t
	var hd C.uint32_t
	r := C.int(0)
	if raw {
		r = 1
	}
	_, err := locked(func() C.int { return C.tm_batch_decode(b.h, r, nil, &n, &hd) })
	return uint64(n), int(hd), err
}
func (b *HipBatch) DecodeTimed(raw bool) (uint64, int, [3]float32, error) {
	var n C.uint64_t
	var hd C.uint32_t
	var ms [3]C.float
	r := C.int(0)
	if raw {
		r = 1
	}
	_, err := locked(func() C.int { return C.tm_batch_decode_timed(b.h, r, nil, &n, &hd, &ms[0]) })
	return uint64(n), int(hd), [3]float32{float32(ms[0]), float32(ms[1]), float32(ms[2])}, err
}
func (b *HipBatch) DecodedText(ndocs int, capacity uint64) ([]byte, []uint64, error) {
	off := make([]uint64, ndocs+1)
	for {
		out := make([]byte, capacity+1)
		rc, err := locked(func() C.int {
			return C.tm_batch_decoded_download(b.h, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(capacity), (*C.uint64_t)(unsafe.Pointer(&off[0])))
		})
		if err != nil {
			return nil, nil, err
		}
		if rc == C.TM_E_NOSPACE {
			capacity = off[ndocs]
			continue
		}
		return out[:off[ndocs]], off, nil
	}
}

// ---- the host normalizer (every flag; what the device hands over to, and what Normalize of go/tokenmonster.go:953 does) ------------------------

func HipNormalize(data []byte, capcode, normFlag uint8) ([]byte, error) {
	if len(data) == 0 {
		return []byte{}, nil
	}
	var out *C.uint8_t
	var n C.size_t
	if _, err := locked(func() C.int {
		return C.tm_normalize((*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)), C.uint32_t(capcode), C.uint32_t(normFlag), &out, &n)
	}); err != nil {
		return nil, err
	}
	defer C.tm_free(unsafe.Pointer(out))
	res := make([]byte, int(n))
	copy(res, unsafe.Slice((*byte)(unsafe.Pointer(out)), int(n))) // (not C.GoBytes: its length is a C int)
	return res, nil
}
func HipNormalizeBatch(docs [][]byte, capcode, normFlag uint8, threads int) ([]byte, []uint64, error) {
	text, offsets := pack(docs)
	outOff := make([]uint64, len(docs)+1)
	var out *C.uint8_t
	if _, err := locked(func() C.int {
		return C.tm_normalize_batch((*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(len(docs)), C.uint32_t(capcode),
			C.uint32_t(normFlag), C.uint32_t(threads), &out, (*C.uint64_t)(unsafe.Pointer(&outOff[0])))
	}); err != nil {
		return nil, nil, err
	}
	defer C.tm_free(unsafe.Pointer(out))
	n := outOff[len(docs)]
	res := make([]byte, n)
	copy(res, unsafe.Slice((*byte)(unsafe.Pointer(out)), int(n)))
	return res, outOff, nil
}

// ---- .tok token dictionaries (training/trainvocab.go:412-480) ----------------------------------------------------------------------------------

// ReadHipTok parses a .tok file: header, tokens, scores (nil if the file has none), special tokens.
func ReadHipTok(file []byte) (header [5]byte, tokens [][]byte, scores []float32, special [][]byte, err error) {
	var blob, sblob *C.uint8_t
	var off, soff *C.uint32_t
	var sc *C.float
	var count, nspecial C.uint32_t
	if _, err = locked(func() C.int {
		return C.tm_tok_read((*C.uint8_t)(unsafe.Pointer(&file[0])), C.size_t(len(file)), (*C.uint8_t)(unsafe.Pointer(&header[0])), &blob, &off, &count, &sc, &sblob, &soff, &nspecial)
	}); err != nil {
		return
	}
	defer func() {
		for _, p := range []unsafe.Pointer{unsafe.Pointer(blob), unsafe.Pointer(off), unsafe.Pointer(sc), unsafe.Pointer(sblob), unsafe.Pointer(soff)} {
			if p != nil {
				C.tm_free(p)
			}
		}
	}()
	cut := func(b *C.uint8_t, o *C.uint32_t, n int) [][]byte {
		res := make([][]byte, n)
		if n == 0 {
			return res
		}
		offs := unsafe.Slice((*uint32)(unsafe.Pointer(o)), n+1)
		bytes := unsafe.Slice((*byte)(unsafe.Pointer(b)), int(offs[n]))
		for i := range res {
			res[i] = append([]byte(nil), bytes[offs[i]:offs[i+1]]...)
		}
		return res
	}
	tokens = cut(blob, off, int(count))
	special = cut(sblob, soff, int(nspecial))
	if sc != nil {
		scores = append([]float32(nil), unsafe.Slice((*float32)(unsafe.Pointer(sc)), int(count))...)
	}
	return
}

// WriteHipTok is the inverse (scores and special may be nil).
func WriteHipTok(header [5]byte, tokens [][]byte, scores []float32, special [][]byte) ([]byte, error) {
	blob, off64 := pack(tokens)
	off := make([]uint32, len(off64))
	for i, o := range off64 {
		off[i] = uint32(o)
	}
	sblob, soff64 := pack(special)
	soff := make([]uint32, len(soff64))
	for i, o := range soff64 {
		soff[i] = uint32(o)
	}
	var sc *C.float
	if len(scores) > 0 {
		sc = (*C.float)(unsafe.Pointer(&scores[0]))
	}
	var out *C.uint8_t
	var n C.size_t
	if _, err := locked(func() C.int {
		return C.tm_tok_write((*C.uint8_t)(unsafe.Pointer(&header[0])), (*C.uint8_t)(unsafe.Pointer(&blob[0])), (*C.uint32_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(tokens)), sc,
			(*C.uint8_t)(unsafe.Pointer(&sblob[0])), (*C.uint32_t)(unsafe.Pointer(&soff[0])), C.uint32_t(len(special)), &out, &n)
	}); err != nil {
		return nil, err
	}
	defer C.tm_free(unsafe.Pointer(out))
	return C.GoBytes(unsafe.Pointer(out), C.int(n)), nil
}

// ---- device blocks: one vocabulary's tables as ONE range of device memory (a replica per GPU of a node without parsing the file again) --------

// ExportBlock describes the device block of the vocabulary; ImportHipBlock makes a vocabulary around a block of that shape on `device`, whose
// bytes the caller then fills (HipDeviceCopy inside one process, an RCCL broadcast between processes) before the first use.
func (hv *HipVocab) ExportBlock() (C.tm_vocab_block, unsafe.Pointer, error) {
	var m C.tm_vocab_block
	var p unsafe.Pointer
	_, err := locked(func() C.int { return C.tm_vocab_block_export(hv.h, &m, &p) })
	return m, p, err
}
func ImportHipBlock(m *C.tm_vocab_block, device int) (*HipVocab, unsafe.Pointer, error) {
	var h *C.tm_vocab
	var p unsafe.Pointer
	if _, err := locked(func() C.int { return C.tm_vocab_block_import(m, C.int(device), &h, &p) }); err != nil {
		return nil, nil, err
	}
	return &HipVocab{h: h, len: int(C.tm_vocab_size(h))}, p, nil
}
func HipDeviceCopy(dst, src unsafe.Pointer, bytes uint64) error {
	_, err := locked(func() C.int { return C.tm_device_copy(dst, src, C.uint64_t(bytes)) })
	return err
}

// ---- several devices: what tokenmonster_hip.go's HipDevices / HipVocabSet / HipDatasetSet leave out --------------------------------------------

// OpenHipDeviceList opens exactly these devices (tm_devices_open takes the first n).
func OpenHipDeviceList(devices []int) (*HipDevices, error) {
	list := make([]C.int, len(devices))
	for i, d := range devices {
		list[i] = C.int(d)
	}
	var g *C.tm_devices
	if _, err := locked(func() C.int { return C.tm_devices_open_list(&list[0], C.int(len(list)), &g) }); err != nil {
		return nil, err
	}
	return &HipDevices{g}, nil
}
func (g *HipDevices) Device(member int) int { return int(C.tm_devices_device(g.h, C.int(member))) }

// RcclRanks: the ranks of the RCCL communicator behind ScoreCandidateAll (0: none - and why).
func (g *HipDevices) RcclRanks() (int, string) {
	var why *C.char
	n := int(C.tm_devices_rccl_ranks(g.h, &why))
	if why != nil {
		return n, C.GoString(why)
	}
	return n, ""
}
func (s *HipVocabSet) Count() int { return int(C.tm_vocab_set_count(s.h)) }

// Tune lays the tables of every replica out by use on a sample of NORMALIZED text (results unchanged; worth it for large vocabularies).
func (s *HipVocabSet) Tune(normalizedSample []byte) error {
	_, err := locked(func() C.int {
		return C.tm_vocab_set_tune(s.h, (*C.uint8_t)(unsafe.Pointer(&normalizedSample[0])), C.uint64_t(len(normalizedSample)))
	})
	return err
}

// Range: the bytes of the dataset member `member` owns, and the bytes of following text it holds beside them.
func (ds *HipDatasetSet) Range(member int) (uint64, uint64) {
	var halo C.uint64_t
	n := uint64(C.tm_dataset_set_range(ds.h, C.int(member), &halo))
	return n, uint64(halo)
}

// ---- the scoring pass without the copy to the host: the histogram stays in HBM (a caller with its own collective, e.g. RCCL through cgo) -------

// ScoreDevice enqueues the pass on the NULL stream and returns where the histogram lies (n_ids + 4 + 256 uint32 words); ScoreDeviceInto copies
// it behind the pass into device memory of the caller.
func ScoreDevice(hv *HipVocab, d *HipDataset, stripOff, stripLen []uint64) (unsafe.Pointer, uint64, error) {
	var so, sl *C.uint64_t
	if len(stripOff) > 0 {
		so = (*C.uint64_t)(unsafe.Pointer(&stripOff[0]))
		sl = (*C.uint64_t)(unsafe.Pointer(&stripLen[0]))
	}
	var hist *C.uint32_t
	var words C.uint64_t
	_, err := locked(func() C.int { return C.tm_score_device(hv.h, d.h, so, sl, C.uint32_t(len(stripOff)), nil, &hist, &words) })
	return unsafe.Pointer(hist), uint64(words), err
}
func ScoreDeviceInto(hv *HipVocab, d *HipDataset, stripOff, stripLen []uint64, dst unsafe.Pointer, dstWords uint64) error {
	var so, sl *C.uint64_t
	if len(stripOff) > 0 {
		so = (*C.uint64_t)(unsafe.Pointer(&stripOff[0]))
		sl = (*C.uint64_t)(unsafe.Pointer(&stripLen[0]))
	}
	_, err := locked(func() C.int {
		return C.tm_score_device_into(hv.h, d.h, so, sl, C.uint32_t(len(stripOff)), nil, (*C.uint32_t)(dst), C.uint64_t(dstWords))
	})
	return err
}

// ---- the streaming decoder on serialized ids (go/tokenmonster.go:640 DecodeSerialized; server jobs 7 - 9) -----------------------------------------

func (d *HipDecoder) DecodeSerialized(data []byte, encodingLength uint8) ([]byte, error) {
	var p *C.uint8_t
	if len(data) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	capBytes := uint64(len(data))*8 + 64
	var n C.uint64_t
	out := make([]byte, capBytes+1)
	rc, err := locked(func() C.int {
		return C.tm_decoder_decode_serialized(d.h, p, C.uint64_t(len(data)), C.uint32_t(encodingLength), (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(capBytes), &n)
	})
	if err != nil {
		return nil, err
	}
	if rc == C.TM_E_NOSPACE { // the ids HAVE been consumed and the text is kept: fetch it with a buffer of the size reported
		out = make([]byte, uint64(n)+1)
		if _, err = locked(func() C.int {
			return C.tm_decoder_decode(d.h, nil, 0, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.uint64_t(n), &n)
		}); err != nil {
			return nil, err
		}
	}
	return out[:n], nil
}

// ---- the entry points that use the calling thread's CURRENT device (HipSetDevice) instead of naming one: for a host that pins a goroutine to
// an OS thread per GPU (runtime.LockOSThread) and keeps it there ---------------------------------------------------------------------------------

func LoadHipCurrent(image []byte) (*HipVocab, error) {
	var h *C.tm_vocab
	if _, err := locked(func() C.int { return C.tm_vocab_load((*C.uint8_t)(unsafe.Pointer(&image[0])), C.size_t(len(image)), &h) }); err != nil {
		return nil, err
	}
	return &HipVocab{h: h, len: int(C.tm_vocab_size(h))}, nil
}

// LoadHipCurrentSample is LoadHipCurrent with the tables laid out by use from the start (tm_vocab_load_sample): normalizedSample is a few MiB of
// the text the vocabulary will be used on, as Normalize writes it.  Results are the same; the match kernel is ~5 % faster with 100 000 ids.
func LoadHipCurrentSample(image []byte, normalizedSample []byte) (*HipVocab, error) {
	var h *C.tm_vocab
	var p *C.uint8_t
	if len(normalizedSample) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&normalizedSample[0]))
	}
	if _, err := locked(func() C.int {
		return C.tm_vocab_load_sample((*C.uint8_t)(unsafe.Pointer(&image[0])), C.size_t(len(image)), p, C.uint64_t(len(normalizedSample)), &h)
	}); err != nil {
		return nil, err
	}
	return &HipVocab{h: h, len: int(C.tm_vocab_size(h))}, nil
}
func UploadDatasetCurrent(normalized []byte) (*HipDataset, error) {
	var d *C.tm_dataset
	if _, err := locked(func() C.int { return C.tm_dataset_upload((*C.uint8_t)(unsafe.Pointer(&normalized[0])), C.uint64_t(len(normalized)), &d) }); err != nil {
		return nil, err
	}
	return &HipDataset{d}, nil
}

// CountNormalizedBatch is Count (go :971) over ALREADY NORMALIZED documents (CountBatch takes raw ones).
func (hv *HipVocab) CountNormalizedBatch(normalized [][]byte) ([]int, []int, error) {
	n := len(normalized)
	text, offsets := pack(normalized)
	counts := make([]uint64, n+1)
	missing := make([]uint32, n+1)
	if _, err := locked(func() C.int {
		return C.tm_count_batch(hv.h, (*C.uint8_t)(unsafe.Pointer(&text[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])), C.uint32_t(n),
			(*C.uint64_t)(unsafe.Pointer(&counts[0])), (*C.uint32_t)(unsafe.Pointer(&missing[0])))
	}); err != nil {
		return nil, nil, err
	}
	res := make([]int, n)
	miss := make([]int, n)
	for i := range res {
		res[i] = int(counts[i])
		miss[i] = int(missing[i])
	}
	return res, miss, nil
}

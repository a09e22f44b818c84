#!/usr/bin/env python3
"""tools/decode_scripts.py [MiB] — development aid: the device-resident decode (tm_batch_decode_timed) on text of other scripts than the bench corpus's:
what k_dec_capcode takes per GiB of encoded text when every 64-byte chunk has characters beyond ASCII."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import _native as N, synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(7)
WORDS = {
    "english": "the quick brown fox jumps over the lazy dog and The Cat sat on THE MAT while it's raining".split(),
    "russian": "и в не на Я что он с как это по но они Мы бы её только ещё было для Москва РОССИЯ который сказал время".split(),
    "greek": "και το να είναι από την με για Ο στο που δεν Η θα ΑΘΗΝΑ τους αυτό έχει πολύ".split(),
    "french": "le la les des été où à ça Français ÉTÉ déjà très même être cœur garçon l’homme aujourd’hui".split(),
    "cjk+emoji": "世界 你好 日本語 東京 😀 🎉 こんにちは テスト hello 漢字 中文 👍".split(),
    "japanese": "これは 日本語 の テキスト です ございます がんばって ください データ プログラム ヴァイオリン 東京 大学 で を つかう ぱぴぷぺぽ ばびぶべぼ Tokyo".split(),
    "hindi": "यह हिन्दी का पाठ है और इसमें कई शब्द हैं जैसे कि विश्वविद्यालय प्रौद्योगिकी स्वतंत्रता क्या क्यों नहीं भारत दिल्ली मुम्बई १२३".split(),
    "thai": "ภาษาไทย อยู่ ที่ นี่ กรุงเทพมหานคร ประเทศไทย สวัสดี ครับ ค่ะ น้ำ ผู้ ใหญ่ ไม่ ได้ เป็น คุณ รู้ เรื่อง ๑๒๓".split(),
    "bengali+tamil": "বাংলাদেশের কোনো হবে বাংলা ভাষা মানুষ কলকাতা தமிழ் மொழி போகிறோம் சென்னை கொண்டு".split(),
    "fr: no U+2019": "le la les des été où à ça Français ÉTÉ déjà très même être cœur garçon homme aujourdhui".split(),
    "fr: accents only": "le la les des été où à ça déjà très même être garçon homme aujourdhui".split(),
    "ascii + U+2019": "le la les des ete ou a ca deja tres meme etre l’homme aujourd’hui".split(),
    "ascii + œ": "le la les des ete ou a ca deja tres meme etre cœur".split(),
    "ascii + É caps": "le la les des ete ou a ca deja tres meme etre ÉTÉ Français".split(),
}
toks = [bytes([c]) for c in range(256)]
v = tm.Vocab(synth.build_vocab(toks, capcode=2, charset=1, norm_flag=1))
for name, words in WORDS.items():
    docs, total = [], 0
    while total < (mb << 20):
        n = int(rng.integers(200, 8000))
        d = " ".join(words[int(i)] for i in rng.integers(0, len(words), size=n // 5)).encode()
        docs.append(d); total += len(d)
    raw, offs = tm.pack_documents(docs)
    nd = offs.size - 1
    b = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(raw.size) * 2 + (1 << 20), nd, C.byref(b)))
    N.check(N.lib.tm_batch_upload_raw(b, N.ptr(raw), N.ptr(offs), nd))
    import time
    N.check(N.lib.tm_batch_normalize(b, None)); N.check(N.lib.tm_batch_run(b, None))
    nt, nm = C.c_uint64(), C.c_uint64()
    N.check(N.lib.tm_batch_totals(b, C.byref(nt), C.byref(nm)))
    t0 = time.perf_counter()
    for _ in range(3):
        N.check(N.lib.tm_batch_normalize(b, None))
    t_norm = (time.perf_counter() - t0) / 3 * 1e3
    t0 = time.perf_counter()
    for _ in range(3):
        N.check(N.lib.tm_batch_run(b, None))
    N.check(N.lib.tm_batch_totals(b, C.byref(nt), C.byref(nm)))
    t_run = (time.perf_counter() - t0) / 3 * 1e3
    enc = int(N.lib.tm_batch_normalized_bytes(b))
    print("%-18s tokenize: normalize %.2f ms + K0-K4 %.2f ms per GiB of raw text (%d documents to the host normalizer)" % (
        name, t_norm * (1 << 30) / raw.size, t_run * (1 << 30) / raw.size, int(N.lib.tm_batch_host_fallback_docs(b))), flush=True)
    nbytes, hostd = C.c_uint64(), C.c_uint32()
    ms = (C.c_float * 3)()
    acc = np.zeros(3)
    for i in range(4):
        N.check(N.lib.tm_batch_decode_timed(b, 0, None, C.byref(nbytes), C.byref(hostd), ms))
        if i: acc += np.array(list(ms))
    acc /= 3
    text = np.frombuffer(b"".join(docs), dtype=np.uint8)
    n64 = text.size // 64 * 64
    hi = float((text[:n64].reshape(-1, 64) >= 0x80).any(axis=1).mean())
    # (per GiB of DECODED text: with this byte-level vocabulary the ids of a character beyond ASCII decode to more bytes than the character had - its
    # single-byte tokens are stored escaped -, so the text the decoder reads is larger than the normalized text by a factor that depends on the script)
    dec = float(nbytes.value)
    print("%-18s %6.1f MB normalized -> %6.1f MB decoded, %3.0f %% of the raw chunks beyond ASCII, host docs %d of %d: capcode %.3f ms = %.2f ms per GiB of decoded text (gather %.3f)" % (
        name, enc / 1e6, dec / 1e6, 100 * hi, hostd.value, nd, acc[2], acc[2] * (1 << 30) / max(dec, 1.0), acc[1]), flush=True)
    N.lib.tm_batch_free(b)

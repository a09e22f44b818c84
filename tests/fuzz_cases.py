"""Differential fuzz cases shared by tests/test_gpu_fuzz.py (-m gpu: a few seeds on the real device, or on the emulated one under
TM_EMU=1) and tools/emu/fuzz.py (as many seeds as time allows on the emulated device): random small vocabularies (capcode 0 / 2,
UTF-8 / UTF-16, with and without an unk token), batches of documents whose lengths sit on and around the segment (256), tile (8 x 256)
and group boundaries, through tokenize / count / serialized / score (strips) / decode against the oracle; and the device normalizer
against the host normalizer on runs of capitals, digits and apostrophes of every length."""
import numpy as np

import conftest
import tokenmonster_amd as tm
from tokenmonster_amd import _native as N, synth
from oracle_bind import Oracle

LENS = [0, 1, 2, 3, 39, 40, 41, 80, 215, 216, 217, 255, 256, 257, 295, 296, 297, 511, 512, 513, 767, 1023, 1024, 1025, 2047, 2048, 2049, 4096]


def one(seed):
    rng = np.random.default_rng(seed)
    capcode = int(rng.choice([0, 2, 2]))
    charset = int(rng.choice([1, 1, 1, 2]))
    unk = bool(rng.random() < 0.3)
    toks = conftest.fuzz_vocab_tokens(rng, capcode, int(rng.integers(20, 400)), singles=bool(rng.random() < 0.85))
    if charset == 2:
        toks = sorted({bytes(b for ch in t for b in (ch, 0))[:40] for t in toks if len(t) <= 20})
    img = synth.build_vocab(toks, capcode=capcode, charset=charset, with_unk=unk)
    v, orc = tm.Vocab(img), Oracle(img)
    nd = int(rng.integers(1, 40))
    docs = []
    for _ in range(nd):
        r = rng.random()
        n = int(rng.choice(LENS)) if r < 0.6 else int(rng.integers(0, 6000)) if r < 0.97 else int(rng.integers(60000, 140000))
        t = conftest.fuzz_text(rng, capcode, max(n, 1))[:n]
        if charset == 2:
            t = bytes(b for ch in t[: n // 2] for b in (ch, 0)) + (b"a" if n % 2 else b"")
        docs.append(t)
    text, offs = tm.pack_documents(docs)
    # (a third of the cases with the tables laid out by use on some of the text, tm_vocab_tune: its own random stream, so that a seed's case
    # is what it was before round 4)
    rng_t = np.random.default_rng(seed ^ 0x5EED)
    if rng_t.random() < 0.33:
        v.tune(text[: int(rng_t.integers(0, text.size + 1))])
    # (a quarter of the cases through the id-staging form of the K4 walk - test hook 15 -, which the two-plane rows otherwise do not take)
    from tokenmonster_amd import _native as N
    old_flags = N.lib.tm_debug_flags(32768 if rng.random() < 0.25 else 0)
    try:
        ids, toff, missing = v.tokenize_packed(text, offs)
    finally:
        N.lib.tm_debug_flags(old_flags)
    counts, cmiss = v.count_packed(text, offs)
    for d, doc in enumerate(docs):
        exp, miss = orc.tokenize(doc)
        got = ids[int(toff[d]):int(toff[d + 1])]
        if got.size != exp.size or (got != exp).any() or int(missing[d]) != miss:
            raise AssertionError("seed %d doc %d (len %d): ids differ" % (seed, d, len(doc)))
        ec, em = orc.count(doc)
        if int(counts[d]) != ec or int(cmiss[d]) != em:
            raise AssertionError("seed %d doc %d: count %d/%d vs %d/%d" % (seed, d, int(counts[d]), int(cmiss[d]), ec, em))
    # serialized, every width the vocabulary allows
    for enc in (0, 2, 3, 4):
        blob, boff, _, e = v.tokenize_serialized_packed(text, offs, encoding_length=enc)
        w = np.zeros((ids.size, 4), dtype=np.uint8)
        w[:, :e] = np.asarray(blob[: ids.size * e]).reshape(ids.size, e)
        if not (w.view(np.uint32).ravel() == ids).all():
            raise AssertionError("seed %d: serialized width %d" % (seed, enc))
    # scoring: the documents concatenated as one dataset, one strip and a few strips
    data = np.ascontiguousarray(text)
    if data.size:
        import ctypes as C
        ds = C.c_void_p()
        N.check(N.lib.tm_dataset_upload(N.ptr(data), int(data.size), C.byref(ds)))
        cuts = sorted(set([0, int(data.size)] + [int(x) for x in rng.integers(0, data.size + 1, size=int(rng.integers(0, 4)))]))
        so = np.array(cuts[:-1], dtype=np.uint64)
        sl = np.array([b - a for a, b in zip(cuts, cuts[1:])], dtype=np.uint64)
        got = np.zeros(v.n_ids(), dtype=np.uint32)
        tit = C.c_uint64()
        ms = np.zeros(32, dtype=np.uint8)
        N.check(N.lib.tm_score(v.handle, ds, N.ptr(so), N.ptr(sl), so.size, N.ptr(got), C.byref(tit), N.ptr(ms)))
        exp_s = np.zeros(v.n_ids(), dtype=np.uint64)
        exp_t, exp_m = 0, np.zeros(32, dtype=np.uint8)
        for a, b in zip(cuts, cuts[1:]):
            s_, t_, m_ = orc.score(data[a:b])
            exp_s += s_
            exp_t += t_
            exp_m |= m_
        N.lib.tm_dataset_free(ds)
        if not ((got == exp_s).all() and tit.value == exp_t and (ms == exp_m).all()):
            raise AssertionError("seed %d: score histogram over strips %s" % (seed, cuts))
    # raw decode = concatenated token bytes
    out, ooff = v.decode_packed(ids, toff, raw=True)
    for d in range(nd):
        exp = orc.decode_raw(ids[int(toff[d]):int(toff[d + 1])])
        if out[int(ooff[d]):int(ooff[d + 1])].tobytes() != exp:
            raise AssertionError("seed %d doc %d: decode_raw" % (seed, d))
    return sum(len(x) for x in docs)


NORM_ALPHABET = [chr(c) for c in b"aabcxyzABCDQWXYZ   ''1239..,-_()\n\t"] + ["\u2019", "\u2019", "\u201c", "\u2014", "\u2026", "\u00e9", "\u00c9", "\u4e2d", "\U0001f600", "\u0301"]


# accented Latin: two-byte characters that decompose under NFD (é -> e + U+0301), that stay whole (Æ ø ß Ł), letters without case (ª º),
# symbols, and the two whose lower case is special (İ ı)
LATIN = [chr(c) for c in (0xE9, 0xC9, 0xE8, 0xC0, 0xEF, 0xF1, 0xD1, 0xFC, 0xDC, 0xE7, 0xC7, 0xC6, 0xE6, 0xD8, 0xF8, 0xDF, 0xDE, 0xFE, 0xD0, 0xF0, 0xAA, 0xBA, 0xB5, 0xAB, 0xBB,
                          0xA0, 0xB2, 0xBD, 0xD7, 0xF7, 0x100, 0x101, 0x10C, 0x10D, 0x141, 0x142, 0x152, 0x153, 0x130, 0x131, 0x149, 0x17F, 0x178, 0xFF, 0x138, 0x13F, 0x140)]


# round 5: what the device pass took over last - four-byte characters (emoji, a skin-tone modifier, plane-2 ideographs, hieroglyphs; Deseret
# with its case, a mathematical letter and a musical symbol that decomposes keep their documents on the host) and Hangul syllables with and
# without a final consonant (NFD: jamo by arithmetic)
ASTRAL_HANGUL = ["\U0001f600", "\U0001f680", "\U0001f44d\U0001f3fd", "\U0001f1e9\U0001f1ea", "\U00020000", "\U00020bb7", "\U00013000", "\U0001d49c", "\U00010400", "\U00010428",
                 "\U0001d15e", "\uac00", "\uac01", "\ud7a3", "\ud55c", "\uad6d", "\uc5b4", "\ubdc1", "\uac12", "\ub2ed", "\u1100", "\u1161", "\u11a8", "\ufe0f"]


# round 6: Latin Extended Additional under NFD - a letter and one or two marks per character: the letters Vietnamese adds to Latin-1 / Latin
# Extended-A, a few of the others of the block, and the ones that stay with the host (a two-byte base: U+1E9B; no decomposition: U+1E9E, U+1EFF)
VIET = list("ếệềểễấậầẩẫắặằẳẵớợờởỡứựừửữạảịỉọỏụủỳỵỷỹẾỆẤẬẮẶỚỢỨỰẠẢỊỌỤỲỸđĐơƠưƯăĂâêô") + ["\u1e00", "\u1e01", "\u1e9b", "\u1e9e", "\u1eff", "\u1e69", "\u1e08"]


# round 6: kana, voiced and not (a voiced kana is a kana and its mark under NFD, on the device), the marks by themselves (the host's), the iteration marks
KANA = list("あかきくけこさしたなはひまやらわんがぎぐげござじずだぢづでどばびぶべぼぱぴぷぺぽゔゞアカサタハガギグザジダヂヅデドバビブベボパピプペポヴヷヸヹヺヾー") + ["\u3099", "\u309a", "\u0301", "\u4e2d"]


# round 6: Devanagari and Thai with their marks of canonical class > 0 (virama 9, nukta 7, the Thai vowels below 103 and tone marks 107: in and out of
# order), three-byte digits, letters that decompose (the host's), a Latin mark among them
INDIC_THAI = list("कखगजडतनपमरलवसहअआइएओ") + ["\u093e", "\u093f", "\u0940", "\u0947", "\u094b", "\u094d", "\u094d", "\u093c", "\u0902", "\u0951", "\u0952", "१", "२", "\u0929", "\u0958"] + \
             list("กขคงจดตนบปมยรลวสหอาเแไ") + ["\u0e31", "\u0e34", "\u0e35", "\u0e38", "\u0e39", "\u0e48", "\u0e49", "\u0e4a", "\u0e4c", "๑", "๒", "\u09cb", "\u0301", "\u09cc", "ক", "\u0bca", "த", "\u0958", "\u095b", "\u0ccb", "\u1026", "ა", "ბ", "ქ", "ᴀ", "Ა"]


def one_norm(seed):
    """the device normalizer (NFD / lowercase + capcode 2; host fallback for what it does not do itself) against the host normalizer:
    runs of capitals, digits and apostrophes of every length, also across the 64-byte chunks and 1 KiB pieces of the device pass"""
    rng = np.random.default_rng(seed)
    capcode, flag = (2, int(rng.choice([1, 1, 3, 0]))) if rng.random() < 0.85 else (0, int(rng.choice([1, 3])))
    # (round 6) every other case: any of the 256 flag values - quotemarks 8, collapse 16, trim 32, leadingspace 64, unixlines 128 and accents 4
    # go through the device's filter pass (tm_norm.hip: k_pf_*) - over text with runs of blanks, CR LF pairs and curly quotes
    lossy = rng.random() < 0.5
    if lossy:
        flag = int(rng.integers(0, 256))
    toks = [bytes([c]) for c in range(256)]
    v = tm.Vocab(synth.build_vocab(toks, capcode=capcode, charset=1, norm_flag=flag))
    docs = []
    for _ in range(int(rng.integers(1, 60))):
        parts = []
        n = int(rng.choice([0, 1, 5, 60, 63, 64, 65, 127, 129, 1000, 1023, 1024, 1025, 2047, 2049, 3100]))
        if lossy and rng.random() < 0.01:
            n = int(rng.choice([16383, 16385, 33000]))       # more than one span of the filter pass (tm_norm.hip: PF_SPAN)
        while sum(len(x) for x in parts) < n:
            r = rng.random()
            if r < 0.15:
                parts.append("".join(rng.choice(NORM_ALPHABET[:30] + LATIN, size=int(rng.integers(1, 40)))))
            elif r < 0.22:
                parts.append("".join(rng.choice(NORM_ALPHABET[:30] + ASTRAL_HANGUL[int(rng.integers(0, 12)):], size=int(rng.integers(1, 40)))))
            elif r < 0.27:
                parts.append("".join(rng.choice(NORM_ALPHABET[:36] + VIET, size=int(rng.integers(1, 40)))))
            elif r < 0.30:
                parts.append("".join(rng.choice(NORM_ALPHABET[:30] + KANA, size=int(rng.integers(1, 40)))))
            elif r < 0.33:
                parts.append("".join(rng.choice(NORM_ALPHABET[:30] + INDIC_THAI, size=int(rng.integers(1, 40)))))
            elif lossy and r < 0.5:
                parts.append("".join(rng.choice([" ", "  ", "   ", "\r\n", "\r", "\n", "\t", "\u2018", "\u2019", "\u201c", "\u201d", "\u2019s", "a", "B", "é", "É", "x ", " y", "\u0301", "ñ", "1"],
                                                size=int(rng.integers(1, 30)))))
            elif r < 0.35:
                parts.append("".join(rng.choice(NORM_ALPHABET[:36], size=int(rng.integers(1, 30)))))
            elif r < 0.55:
                # (runs of digits / apostrophes without a capital, longer than the 64-byte margins of the one-pass normalizer: the exact path)
                parts.append(str(rng.choice(["A", "Q", "AB", "Ab", "I'M", "X\u2019S", "A1", "1A", "a'B", "7", "'", "12'", "É", "é1"])) * int(rng.integers(1, 400)))
            elif r < 0.75:
                parts.append("x" * int(rng.integers(1, 1100)))
            elif r < 0.95:
                parts.append("".join(rng.choice(NORM_ALPHABET[:41], size=int(rng.integers(1, 12)))))
            else:
                parts.append("".join(rng.choice(NORM_ALPHABET, size=int(rng.integers(1, 8)))))
        docs.append("".join(parts).encode()[: max(n, 0) + int(rng.integers(0, 40))])
    if rng.random() < 0.2:
        docs.append(bytes(rng.integers(0, 256, size=int(rng.integers(1, 300)), dtype=np.uint8)))      # not UTF-8 at all
    raw, offs = tm.pack_documents(docs)
    got, goff, _ = v.normalize_packed_device(raw, offs)
    exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
    if not ((goff == eoff).all() and got.size == exp.size and (got == exp).all()):
        bad = [d for d in range(len(docs)) if int(goff[d + 1] - goff[d]) != int(eoff[d + 1] - eoff[d]) or
               got[int(goff[d]):int(goff[d + 1])].tobytes() != exp[int(eoff[d]):int(eoff[d + 1])].tobytes()]
        raise AssertionError("seed %d: device-normalized text differs from the host normalizer in documents %s (capcode %d flag %d): %r" % (
            seed, bad[:5], capcode, flag, docs[bad[0]][:200] if bad else None))
    return int(raw.size)


def one_raw(seed):
    """raw text -> ids in ONE device pass (tm_batch_upload_raw + tm_batch_normalize + tm_batch_run: the normalized text stays in the
    normalizer's slabs and the match kernel stages its segments from there; with test hook 11 it is packed first; through the chunked
    host-to-host pipeline) against the host normalizer + the oracle's walk: random vocabularies over what the normalizer writes (capcode
    markers included), documents whose pieces (1 KiB of raw text) and segments (256 normalized bytes) fall everywhere relative to each other"""
    rng = np.random.default_rng(seed)
    flag = int(rng.choice([1, 1, 3]))
    toks = conftest.fuzz_vocab_tokens(rng, 2, int(rng.integers(40, 500)), singles=True)
    toks = sorted(set(toks) | {"\u0301".encode(), "\u00e9".encode(), "\u2019".encode(), b"C", b"W", b"D", b"Ca", b"W x", b" the", b"ing ", b"D a"})
    img = synth.build_vocab(toks, capcode=2, charset=1, norm_flag=flag, with_unk=bool(rng.random() < 0.5))
    v, orc = tm.Vocab(img), Oracle(img)
    words = ["the", "The", "THE", "a", "I", "I'm", "X\u2019s", "caf\u00e9", "\u00c9cole", "HTTPServer", "x1", "42", "3.14", "snake_case", "Q", "\u4e2d\u6587", "\u0434\u0430", "\u0394"]
    docs = []
    for _ in range(int(rng.integers(1, 40))):
        n = int(rng.choice([0, 1, 3, 255, 256, 257, 351, 352, 353, 1000, 1023, 1024, 1025, 1279, 1281, 2047, 2048, 2049, 3000, 4097, 9000])) + int(rng.integers(0, 3))
        parts, size = [], 0
        while size < n:
            r = rng.random()
            w = str(rng.choice(words)) if r < 0.7 else "".join(rng.choice(NORM_ALPHABET[:36], size=int(rng.integers(1, 12)))) if r < 0.9 else "x" * int(rng.integers(1, 700))
            sep = str(rng.choice([" ", " ", " ", "\n", ", ", "", "-"]))
            parts.append(w + sep)
            size += len((w + sep).encode())
        docs.append("".join(parts).encode()[:n])
    if rng.random() < 0.3:
        docs.append(("\U0001f600 \u1e9e " * int(rng.integers(1, 30))).encode())                  # a document for the host normalizer: the batch is packed
    hook = 2048 if rng.random() < 0.25 else 0
    old = N.lib.tm_debug_flags(hook)
    try:
        got = v.tokenize(docs)
    finally:
        N.lib.tm_debug_flags(old)
    norm = [v.normalize(d) for d in docs]
    for d, doc in enumerate(norm):
        exp, _ = orc.tokenize(doc)
        if got[d].size != exp.size or (got[d] != exp).any():
            raise AssertionError("seed %d (hook %d) doc %d (raw %d bytes, normalized %d): ids of the one-pass device path differ" % (seed, hook, d, len(docs[d]), len(doc)))
    raw, offs = tm.pack_documents(docs)
    blob, boff, bmiss, enc, st = v.tokenize_pipeline(raw, offs, raw=True, chunk_bytes=int(rng.choice([4096, 20000, 1 << 20])), lanes=int(rng.integers(1, 4)))
    ids = np.concatenate([g for g in got]) if got else np.zeros(0, np.uint32)
    w = np.zeros((ids.size, 4), dtype=np.uint8)
    w[:, :enc] = np.asarray(blob[: ids.size * enc]).reshape(ids.size, enc)
    if int(boff[-1]) != ids.size * enc or not (w.view(np.uint32).ravel() == ids).all():
        raise AssertionError("seed %d: ids of the host-to-host pipeline differ" % seed)
    return int(raw.size)


DEC_ALPHABET = list(b"CCWWDD    aabcxyzQZ019''.,-\n\t_")
# round 4: the two-byte scripts (case pairs that keep or change the lead byte: а/А р/Р, Greek with the final sigma and the letters whose
# upper-case form has another length: ΐ ŉ), Hebrew / Arabic with points and digits, CJK, kana, and what stays with the host (Hangul is
# caseless and decoded on the device too; cased three-byte letters and four-byte characters are not)
DEC_SCRIPTS = list("абвгджзийклмнопрстуфхцчшщъыьэюяёАБРСЯЁѐїλμνξοπρστυφχψωςάώΐΰΑΩΣשלוםְִّمرحبا١٢٣中文字测试。、こんにちはカタガギ한국ḁẞ①→★\U0001F600") + ["\U0001f680", "\U0001f44d\U0001f3fd", "\U00020000", "\U00013000", "\U0001d49c", "\U00010428", "\U0001d7d8", "\uac01", "\u1100\u1161\u11a8"]


# what the device decoder takes on itself beyond ASCII, and what it must hand to the host (upper-case forms of another length or lead byte: \u00ff \u00b5 \u0131 \u017f \u0149)
DEC_LATIN = [chr(c) for c in (0xE9, 0xC9, 0xE8, 0xEF, 0xF1, 0xFC, 0xDC, 0xE7, 0xC6, 0xE6, 0xD8, 0xF8, 0xDF, 0xFE, 0xF0, 0xAA, 0xBA, 0xB5, 0xAB, 0xA0, 0xB2, 0xBD, 0xD7, 0xF7, 0x101, 0x10D, 0x141, 0x142,
                              0x153, 0x130, 0x131, 0x149, 0x17F, 0x17E, 0xFF, 0x138, 0x140, 0x161, 0x151)]


def one_decode(seed):
    """tm_decode_batch (capcode decoding of the pure-ASCII documents on the device, k_dec_capcode; the others on the host) against the
    streaming host decoder (tm_decoder_*), on marker soups no encoder would write: every order of 'C', 'W', 'D', spaces and characters,
    runs across the 64-byte chunks of the device pass, documents that end inside a pending flag."""
    rng = np.random.default_rng(seed)
    # 256 single-byte tokens built WITHOUT capcode (no "D "+token twins, so that reverse[id] is the byte itself), header switched to capcode 2
    img = synth.build_vocab([bytes([c]) for c in range(256)], capcode=0, charset=1)
    v = tm.Vocab(b"\x02" + bytes(img[1:]))
    assert v.capcode() == 2
    docs = []
    for _ in range(int(rng.integers(1, 50))):
        n = int(rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 200, 640, 1000]))
        n += int(rng.integers(0, 5))
        r = rng.random()
        if r < 0.6:
            doc = bytes(rng.choice(DEC_ALPHABET, size=n).tolist())
        elif r < 0.8:
            doc = (bytes(rng.choice(list(b"CWD"), size=int(rng.integers(1, 70))).tolist()) + bytes(rng.choice(DEC_ALPHABET, size=5).tolist())) * (n // 8 + 1)
            doc = doc[:n]
        elif r < 0.92:
            doc = bytes(rng.choice(list(b"W helo wrd's 12"), size=n).tolist())
        elif r < 0.96:
            doc = bytes(rng.choice(DEC_ALPHABET, size=n).tolist()) + "\u00e9\u2019 W\u00e9".encode() + bytes(rng.choice(DEC_ALPHABET, size=7).tolist())
        else:
            doc = bytes(rng.choice(DEC_ALPHABET, size=n).tolist()) + "\u4e2d W\u0416".encode() + bytes(rng.choice(DEC_ALPHABET, size=7).tolist())      # other scripts: the host decoder
        # a third of the documents: accented Latin, decomposed (marks) and not, curly quotes, and now and then a byte that breaks the UTF-8
        if rng.random() < 0.35:
            pieces = []
            for _ in range(max(1, n // 3)):
                q = rng.random()
                if q < 0.45: pieces.append(bytes([int(rng.choice(DEC_ALPHABET))]))
                elif q < 0.60: pieces.append(str(rng.choice(DEC_LATIN)).encode())
                elif q < 0.75: pieces.append(str(rng.choice(DEC_SCRIPTS)).encode())
                elif q < 0.85: pieces.append(bytes([int(rng.choice(list(b"aeoun")))]) + str(rng.choice(["\u0301", "\u0300", "\u0308", "\u0303", "\u0327", "\u036f"])).encode())
                elif q < 0.93: pieces.append(str(rng.choice(["\u2019", "\u2018", "\u201c", "\u2014", "\u2026", "\u2009"])).encode())
                elif q < 0.97: pieces.append(bytes(rng.choice(list(b"CWD "), size=int(rng.integers(1, 4))).tolist()))
                else: pieces.append(bytes([int(rng.choice([0xC3, 0xA9, 0xE2, 0x80, 0xCD, 0xB5, 0xFF, 0xC5]))]))
            doc = b"".join(pieces) + b"."       # (ends in ASCII: the streaming decoder this is compared with holds back what looks like an incomplete character at the end, tokenmonster.cpp:105-107)
        docs.append(doc)
    # the id of every single-byte token, so that the decoder's input IS the document, byte for byte
    all_ids = np.arange(v.n_ids(), dtype=np.uint32)
    rb, ro = v.decode_packed(all_ids, np.arange(v.n_ids() + 1, dtype=np.uint64), raw=True)
    id_of = np.zeros(256, dtype=np.uint32)
    for i in range(v.n_ids()):
        if int(ro[i + 1] - ro[i]) == 1:
            id_of[int(rb[int(ro[i])])] = i
    text, toff = tm.pack_documents(docs)
    tok = id_of[text]
    assert v.decode_packed(tok, toff, raw=True)[0].tobytes() == text.tobytes()
    out, ooff = v.decode_packed(tok, toff, raw=False)
    for d, doc in enumerate(docs):
        dec = v.decoder()
        exp = dec.decode(tok[int(toff[d]):int(toff[d + 1])]) + dec.flush()
        got = out[int(ooff[d]):int(ooff[d + 1])].tobytes()
        if got != exp:
            raise AssertionError("seed %d doc %d: device capcode decode differs from the host decoder\n in  %r\n exp %r\n got %r" % (seed, d, doc[:120], exp[:120], got[:120]))
    return int(text.size)

"""TEST INFRASTRUCTURE: the trainvocab histogram rebuilt from the REFERENCE runtime's id stream.

The reference's C++ runtime (oracle/_ref) has no scoring mode; the scoring accumulation (training/trainvocab.go:1105-1174) is the
same walk with `scores[id] += bytes the token advanced` in place of every emit.  So the histogram follows from what the reference
DOES produce - the ids of the walk over the same text - once every id is given back the number of bytes it advanced, and that is
fixed by the text: at position i an emitted id stands for its key, for the "D "-prefixed duplicate of its key (go/tokenmonster.go:3450-3462:
same id), or - right after a forward-delete marker (the delete token emitted by a b-branch, go :1241-1261, which itself advances by 0)
- for its space-prefixed key minus the prefix (go :1088-1095).  A character without a token emits nothing (or the unk id) and advances
by 1 (go :1269-1276).  Where more than one reading matches the text the replay tries them in turn and keeps the one under which the
whole stream consumes the whole text (depth-first; the readings disagree within a token or two); a delete token is only read as a
marker where the text admits a forward-delete match at all (longest match of ' ' + text more than one byte longer than that of the
text, go :1092).  Nothing of the walk's scoring or selection is restated here: which token was chosen comes from the reference, how
many bytes it covered from the text and the key set."""
import numpy as np


def vocab_keys(img):
    """.vocab image -> ({id: [key bytes, ...]}, header dict)"""
    img = bytes(img)
    n_info = int.from_bytes(img[17:20], "little")
    pos, by_id = 24, {}
    for _ in range(n_info):
        kl = img[pos]
        key = img[pos + 1: pos + 1 + kl]
        p = pos + 1 + kl
        by_id.setdefault(int.from_bytes(img[p + 8:p + 11], "little"), []).append(key)
        pos = p + 15
    hdr = {"capcode": img[0], "charset": img[1], "unk": int.from_bytes(img[8:11], "little"), "n_ids": int.from_bytes(img[14:17], "little"),
           "delete": int.from_bytes(img[20:23], "little")}
    return by_id, hdr


def histogram_from_ids(img, text, ids, missing):
    """-> (scores u32[n_ids], tokens_in_text, missing_set u8[32]) of the walk whose emitted ids are `ids` over `text`, with `missing`
    characters that had no token.  Raises if no reading of the ids consumes the text."""
    by_id, hdr = vocab_keys(img)
    text = bytes(text)
    ids = [int(x) for x in ids]
    n, T = len(text), len(ids)
    NONE = 0xFFFFFF
    delete_id, unk = hdr["delete"], hdr["unk"]
    off = 2 if hdr["charset"] == 2 else 1
    prefix = b" \x00" if off == 2 else b" "

    allkeys = set()
    for ks in by_id.values():
        allkeys.update(ks)
    maxlen = max((len(key) for key in allkeys), default=0)

    def longest(buf):
        """pansearch LongestSubstring on the key set: length of the longest key that is a prefix of buf (0: none)"""
        for L in range(min(len(buf), maxlen), 0, -1):
            if buf[:L] in allkeys:
                return L
        return 0

    def marker_possible(i):
        """a b-branch can only have been taken towards position i if the text there allows it (go :1088-1093): the longest match of
        ' ' + text[i:] is more than one byte longer than the longest match of text[i:] - a fact of the text and the key set"""
        plain = longest(text[i:i + maxlen])
        spaced = longest(prefix + text[i:i + maxlen - off])
        return plain > 0 and spaced > plain + 1

    def options(k, i, fd, miss_left):
        """readings of token k at text position i -> [(advance, fd', consumes a token, missing byte or None)]"""
        out = []
        if k < T:
            x = ids[k]
            if fd:
                for key in by_id.get(x, ()):
                    if key.startswith(prefix) and text.startswith(key[off:], i):      # (advance 0 is possible: the one-byte alternative " " of a space-prefixed match, go :1117-1118)
                        out.append((len(key) - off, 0, True, None))
                return out
            for key in sorted(by_id.get(x, ()), key=len, reverse=True):
                if text.startswith(key, i):
                    out.append((len(key), 0, True, None))
            if x == delete_id and delete_id != NONE and k > 0 and marker_possible(i):
                out.append((0, 1, True, None))                      # the marker of a b-branch: nothing of the text is covered
            if x == unk and unk != NONE and i < n:
                out.append((1, 0, True, text[i]))                   # go :1269-1276 with an unk token
        if not fd and unk == NONE and miss_left > 0 and i < n:
            out.append((1, 0, False, text[i]))                      # ... without one: nothing was emitted for this byte
        return out

    # depth-first over the readings (iterative): a frame = [k, i, fd, missing left, its options, the next option to try]; `path` = the options taken
    path = []
    frames = []
    dead = set()
    k, i, fd, ml = 0, 0, 0, int(missing)
    frames.append([k, i, fd, ml, options(k, i, fd, ml), 0])
    while frames:
        fr = frames[-1]
        k, i, fd, ml = fr[0], fr[1], fr[2], fr[3]
        if k == T and i == n and not fd and (unk != NONE or ml == 0):
            break
        if fr[5] >= len(fr[4]):
            dead.add((k, i, fd, ml))                # no reading from this state on consumes the text: never enter it again
            frames.pop()
            if path:
                path.pop()
            continue
        adv, fd2, tok, mb = fr[4][fr[5]]
        fr[5] += 1
        k2, i2, ml2 = k + (1 if tok else 0), i + adv, ml - (0 if tok or mb is None else 1)
        if i2 > n or (k2, i2, fd2, ml2) in dead:
            continue
        path.append((ids[k] if tok else None, adv, fd2, mb))
        frames.append([k2, i2, fd2, ml2, options(k2, i2, fd2, ml2), 0])
    if not frames:
        raise AssertionError("no reading of the reference's ids consumes the text")
    scores = np.zeros(hdr["n_ids"], dtype=np.uint32)
    missing_set = np.zeros(32, dtype=np.uint8)
    tokens = 0
    for x, adv, fd2, mb in path:
        tokens += 1                                                 # trainvocab.go: tokensInText++ per emitted token, per delete marker (+= 2 on a b-branch) and per missing byte (:1169)
        if mb is not None:
            missing_set[mb >> 3] |= np.uint8(1 << (mb & 7))
            continue
        if fd2:
            scores[x] += 1                                          # scores[deleteToken]++ (:1134, :1143, :1152)
        else:
            scores[x] += adv                                        # scores[id] += bytes covered (:1109 .. :1162)
    return scores, tokens, missing_set

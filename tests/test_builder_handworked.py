"""CPU: hand-worked vectors for the alternatives search of the vocabulary builder (tm_build_vocab), derived on paper from
go/tokenmonster.go:3597-3772 line by line.  The builder cannot be pinned against the Go implementation in this image (no Go
toolchain; go_oracle/ holds the dump program for whoever has one, and tests/test_go_fixtures.py consumes its output when present),
so these vectors are the next best thing: each is small enough to follow with a pencil, and the derivation is in the docstring.

The loop of go :3597 runs `length` from len(token)-1 down to minAltSize over the prefixes of the token that are keys themselves and
fills two slots (index/length/priority1, index2/length2/priority2).  A rule of priority P is offered to slot 1 when
`priority1 < priority2 || (priority1 == priority2 && length <= length2)` (go :3606 and every rule after it) and to slot 2 otherwise, and
is taken only if the slot's priority is < P.  Afterwards the slots are swapped if slot 2 is the better one (go :3766-3769)."""
from test_builder_normalizer import records
from tokenmonster_amd import synth

NONE = 0xFFFFFF


def alts(img, token):
    recs, keys, _, _ = records(img)
    _, _, i1, i2, _ = recs[token]
    return (keys[i1] if i1 != NONE else None, keys[i2] if i2 != NONE else None)


def singles(s):
    return [bytes([c]) for c in sorted(set(s))]


def test_priority_10_then_everything_else():
    """token " hello world" (capcode 2), prefixes in the vocabulary: " hello", " he" (" " alone is below minAltSize).
    go :3520-3526: begins with ' ' + letter -> nWords = 1, minAltSize = 2; :3553 counts the second word -> nWords = 2, so :3577 leaves
    minAltSize at 2.
    length 6 " hello": length <= len-2 and token[6] == ' ' and the rune behind it is a letter (go :3602-3605) -> priority 10; both
      priorities are 0 and both lengths 0 -> slot 1: index = " hello", length 6, priority1 = 10; continue.
    length 3 " he": token[3] = 'l': not the space rule; last rune 'e', next rune 'l': none of letter|non-letter, number|non-number,
      space|non-space, non-space|space, |capcode (go :3650-3716) applies; not the suffix position; "everything else" (go :3736):
      priority1 (10) is neither below nor equal to priority2 (0) -> slot 2: index2 = " he", length2 3, priority2 = 1.
    go :3766: priority2 (1) < priority1 (10): no swap."""
    img = synth.build_vocab(singles(b" helowrdD") + [b" he", b" hello", b" hello world"], capcode=2, charset=1)
    assert alts(img, b" hello world") == (b" hello", b" he")


def test_space_rules_and_the_swap():
    """token "foo bar" (capcode 2), prefixes in the vocabulary: "foo ", "foo", "fo", "f".  minAltSize = 1 (begins with a letter).
    length 4 "foo ": token[4] = 'b' is not a space, so not the space-then-letter rule; last rune ' ', next rune 'b': space|non-space
      (go :3680) -> priority 7, slot 1 (0 == 0, 0 <= 0): index = "foo ", length 4, priority1 = 7.
    length 3 "foo": length <= len-2, token[3] == ' ' and token[4] = 'b' is a letter -> priority 10 (go :3602); priority1 (7) is neither
      below nor equal to priority2 (0) -> slot 2: index2 = "foo", length2 3, priority2 = 10.
    length 2 "fo", length 1 "f": "everything else"; priority1 (7) < priority2 (10) -> slot 1, but priority1 is not < 1: nothing.
    go :3766: priority2 (10) > priority1 (7) -> swapped: index = "foo", index2 = "foo "."""
    img = synth.build_vocab(singles(b" fobarD") + [b"fo", b"foo", b"foo ", b"foo bar"], capcode=2, charset=1)
    assert alts(img, b"foo bar") == (b"foo", b"foo ")


def test_the_suffix_position_is_taken_by_letter_nonletter():
    """token "bob's" (capcode 2), prefixes in the vocabulary: "bob'", "bob", "bo", "b".  hasSuffixPos (go :287-299) = 3: the token ends
    in "'s" and the rune before it is a letter.
    length 4 "bob'": last rune '\\'', next rune 's': no boundary rule applies, 4 != hasSuffix; "everything else" -> slot 1:
      index = "bob'", length 4, priority1 = 1.
    length 3 "bob": last rune 'b' is a letter and the next rune '\\'' is neither a letter nor '_': letter|non-letter (go :3652)
      -> priority 9 and `continue`; priority1 (1) vs priority2 (0): slot 2: index2 = "bob", length2 3, priority2 = 9.
      The "Suffix" rule of go :3719-3734 (priority 8, followed by `break`: quirk Q7) sits BEHIND that switch: at length == hasSuffix the
      rune before the cut is a letter (hasSuffixPos demands it) and the rune after it is the apostrophe of the suffix, so
      letter|non-letter always fires first — the rule and its `break` are unreachable for the two suffixes of go :3157 ("'s", "’s").
    length 2 "bo", 1 "b": "everything else" offered to slot 1 (1 < 9), whose priority is not < 1: nothing.
    go :3766: priority2 (9) > priority1 (1) -> swapped: index = "bob", index2 = "bob'"."""
    img = synth.build_vocab(singles(b" bo's") + [b"D", b"bo", b"bob", b"bob'", b"bob's"], capcode=2, charset=1)
    assert alts(img, b"bob's") == (b"bob", b"bob'")
    # the same through the typographic apostrophe (U+2019, E2 80 99): cut after "bob" again
    t = "bob’s".encode()
    img = synth.build_vocab(singles(b" bos") + [b"\xe2", b"\x80", b"\x99", b"D", b"bob", t], capcode=2, charset=1)
    assert alts(img, t)[0] == b"bob"


def test_capcode0_underscore_is_part_of_the_word():
    """token "foo_bar" (capcode 0, where go :3622-3646 adds the rules non-letter|letter and non-number|number; '_' counts as a letter on
    both sides of every letter rule), prefixes in the vocabulary: "foo_", "foo".
    length 4 "foo_": last rune '_', next rune 'b': non-letter|letter needs `r != '_'`: no; letter|non-letter needs a non-letter
      behind: no -> "everything else": slot 1: index = "foo_", length 4, priority1 = 1.
    length 3 "foo": last rune 'o', next rune '_': letter|non-letter needs `r2 != '_'`: no -> "everything else"; priority1 (1) vs
      priority2 (0) -> slot 2: index2 = "foo", length2 3, priority2 = 1.
    go :3766: equal priorities, length2 (3) < length (4): no swap."""
    img = synth.build_vocab([bytes([c]) for c in range(32, 127)] + [b"foo", b"foo_", b"_bar", b"bar", b"foo_bar"], capcode=0, charset=1)
    assert alts(img, b"foo_bar") == (b"foo_", b"foo")

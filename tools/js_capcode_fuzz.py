"""tools/js_capcode_fuzz.py <file.json> — the host normalizer (tm_normalize: NFD + capcode level 2) against the reference's JavaScript capcode on the
strings tools/js_capcode_fuzz.js wrote:   node tools/js_capcode_fuzz.js 777 30000 > /tmp/f.json && python tools/js_capcode_fuzz.py /tmp/f.json"""
import sys, json, base64, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tokenmonster_amd import synth
cases = json.load(open(sys.argv[1]))
bad = 0; chars = collections.Counter()
for c in cases:
    s = base64.b64decode(c['s']); e = base64.b64decode(c['e'])
    try: g = synth.normalize(s, 2, 1)
    except Exception as ex: g = b'<err %s>' % str(ex).encode()
    if g != e:
        bad += 1
        if bad <= 12: print(repr(s.decode()), "\n   js  ", e, "\n   host", g)
        for ch in s.decode():
            if ord(ch) > 127: chars[ch] += 1
print(len(cases), "cases,", bad, "differ")
print(chars.most_common(40))

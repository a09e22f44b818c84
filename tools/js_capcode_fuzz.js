// tools/js_capcode_fuzz.js <seed> <count> — development aid, THIS container only (node + /root/reference): random strings with characters
// of many scripts (two-, three- and four-byte) between ASCII capitals, digits, apostrophes and spaces, through the reference's OWN capcode
// encoder (javascript/tokenmonster.js from its "// ---- capcode.js ----" marker on, evaluated where it lies; nothing is copied).  Prints
// JSON [{s: base64 of the string, e: base64 of capcode_encode(NFD(s))}]; tools/js_capcode_fuzz.py compares the host normalizer with it.
// (Round 5: 150 000 strings, 0 differences - after the schedule checker had found the capitals without a lower-case form.)
'use strict';
const fs = require('fs'); const vm = require('vm');
const REF = '/root/reference/javascript/tokenmonster.js';
const src = fs.readFileSync(REF, 'utf8'); const at = src.indexOf('// ---- capcode.js ----');
const ctx = {}; vm.createContext(ctx);
vm.runInContext(src.slice(at) + '\nthis.capcode_encode = capcode_encode;', ctx, { filename: REF });
let state = parseInt(process.argv[2] || '12345') >>> 0;
function rnd(n) { state ^= state << 13; state >>>= 0; state ^= state >>> 17; state ^= state << 5; state >>>= 0; return state % n; }
const N = parseInt(process.argv[3] || '20000');
const ascii = "abcxyzABCXYZ0189' .,-\n";
// ranges of code points to draw "exotic" characters from (letters of many scripts; symbols)
const ranges = [[0x80,0x24F],[0x250,0x36F],[0x370,0x3FF],[0x400,0x52F],[0x530,0x58F],[0x590,0x6FF],[0x900,0x97F],[0xE00,0xE7F],[0x10A0,0x10FF],[0x1E00,0x1FFF],[0x2000,0x206F],[0x2100,0x214F],[0x2C00,0x2DFF],[0x3040,0x30FF],[0x4E00,0x4E80],[0xA640,0xA69F],[0xA720,0xA7FF],[0xAC00,0xAC80],[0xFB00,0xFB4F],[0xFF00,0xFFEF],[0x10400,0x1044F],[0x1D400,0x1D4FF],[0x1F600,0x1F64F]];
const out = [];
for (let k = 0; k < N; k++) {
  const r = ranges[rnd(ranges.length)];
  const len = 1 + rnd(14);
  let s = '';
  for (let i = 0; i < len; i++) {
    if (rnd(3) === 0) { const cp = r[0] + rnd(r[1] - r[0] + 1); if (cp >= 0xD800 && cp <= 0xDFFF) continue; s += String.fromCodePoint(cp); }
    else s += ascii[rnd(ascii.length)];
  }
  const nfd = s.normalize('NFD');
  let enc;
  try { enc = ctx.capcode_encode(nfd); } catch (e) { continue; }
  out.push({ s: Buffer.from(s, 'utf8').toString('base64'), e: Buffer.from(enc, 'utf8').toString('base64') });
}
process.stdout.write(JSON.stringify(out));

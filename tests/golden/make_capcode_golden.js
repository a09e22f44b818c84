// make_capcode_golden.js — pins capcode level 2 to the reference's OWN statement of it.
//
//   node tests/golden/make_capcode_golden.js > /tmp/capcode_js.json && gzip -9n -c /tmp/capcode_js.json > tests/golden/capcode_js.json.gz
//
// Evaluates /root/reference/javascript/tokenmonster.js from its "// ---- capcode.js ----" marker on, WHERE IT LIES
// (nothing is copied into this repository), and runs the reference's capcode_encode (:900-1005) and CapcodeDecoder
// (:1007-1065) on a seeded set of strings.  Runs in this container only (node 12 + /root/reference); the JSON it
// prints is the committed fixture the CPU and -m gpu tests replay.
//
// Per string: in = UTF-8 of the raw string, nfd = UTF-8 of String.prototype.normalize('NFD') (what the pipeline feeds
// capcode after go/tokenmonster.go:243), enc = capcode_encode(nfd), dec = CapcodeDecoder.decode(enc).  All base64.
'use strict';
const fs = require('fs');
const vm = require('vm');

const REF = '/root/reference/javascript/tokenmonster.js';
const src = fs.readFileSync(REF, 'utf8');
const at = src.indexOf('// ---- capcode.js ----');
if (at < 0) throw new Error('capcode section not found in ' + REF);
const ctx = {};
vm.createContext(ctx);
vm.runInContext(src.slice(at) + '\nthis.capcode_encode = capcode_encode; this.CapcodeDecoder = CapcodeDecoder;', ctx, { filename: REF });

// xorshift32, seeded
let state = 0x43415043;
function rnd(n) { state ^= state << 13; state >>>= 0; state ^= state >>> 17; state ^= state << 5; state >>>= 0; return state % n; }
function pick(a) { return a[rnd(a.length)]; }

const lower = 'abcdefghijklmnopqrstuvwxyz'.split('');
const upper = 'ABCDEFGHIJKLMNOPQRSTUVWXYZ'.split('');
const digits = '0123456789'.split('');
const punct = ' .,;:-_()[]{}<>/\\"!?@#$%^&*+=|~`\n\t'.split('');
const apos = ["'", '’'];
const gpunct = ['‘', '“', '”', '—', '–', '…', '•', ' ', ' '];
const marks = ['́', '̀', '̈', '̧', '̃'];                 // combining marks (\p{M})
const accented = ['é', 'É', 'ü', 'Ü', 'ñ', 'Ñ', 'ç', 'Ç', 'å', 'Å', 'ö', 'Ö', 'ß', 'ø', 'Ø', 'š', 'Š', 'ž', 'Ž', 'ő', 'Ő'];
// Latin-1 Supplement / Latin Extended-A beyond the accented letters: letters that do not decompose, letters without case, symbols, the
// characters whose case mapping changes the length or the lead byte (what the device normalizer / decoder take or hand to the host)
const latin1 = ['ø', 'Ø', 'æ', 'Æ', 'ð', 'Ð', 'þ', 'Þ', 'ß', 'œ', 'Œ', 'ł', 'Ł', 'đ', 'Đ', 'ħ', 'Ħ', 'ª', 'º', 'µ', '×', '÷', '«', '»', '¿', '¡', '½', '²', '\u00a0', 'ÿ', 'Ÿ', 'ı', 'İ', 'ŉ', 'ſ', 'ĸ', 'ŀ', 'Ŀ', 'à', 'À', 'ê', 'Ê', 'î', 'Î', 'õ', 'Õ', 'ů', 'Ů', 'ż', 'Ż', 'ę', 'Ę'];
const other = ['中', '文', '日', '本', 'あ', 'カ', '한', 'д', 'Д', 'ж', 'Ж', 'λ', 'Λ', 'ω', 'Ω', 'ا', 'ב', '😀', '🚀', 'ǅ', 'ʰ', '٣', '५', '½', 'Ⅷ', 'ª'];
const words = ['the', 'HTTP', 'Server', 'iPhone', 'McDonald', 'NASA', 'it', 'don', 't', 's', 'I', 'M', 'x86', 'USA', 'e', 'Go', 'API', 'v2', 'a', 'B'];

// alphabets by flavour; every flavour keeps the characters capcode's state machine looks at (caps, digits, apostrophes, marks)
const flavours = [
  () => pick([lower, upper, digits, punct, apos, [' '], [' ']]),                                   // ASCII, all classes
  () => pick([lower, upper, upper, [' '], apos, digits]),                                            // capital heavy
  () => pick([lower, upper, digits, punct, apos, gpunct, [' ']]),                                    // + general punctuation (NFD stable)
  () => pick([lower, upper, marks, apos, [' '], accented, digits]),                                  // marks and accents (NFD changes these)
  () => pick([lower, upper, other, digits, apos, [' '], punct, marks]),                              // CJK, Cyrillic, Greek, emoji, titlecase, other digits
  () => pick([words, words, [' '], [' '], apos, punct, digits, upper]),                              // word pieces
  () => pick([lower, upper, latin1, latin1, accented, apos, [' '], digits, marks, gpunct]),            // Latin-1 / Latin Extended-A heavy
];

const inputs = [];
// the strings tests/test_gpu_parity.py feeds the device normalizer by hand
['', 'A', 'a', 'AB', 'Ab', 'aB', 'ABc', 'ABC', ' ABC d', 'HTTPServer2Go x', "X's Y'S it's 'a' I'M", '12AB34cd', 'A1B2c', 'X’s Y’S it’s',
 'A'.repeat(200) + 'b', 'A'.repeat(200), 'a' + 'B'.repeat(130) + ' ' + 'C'.repeat(70) + 'd', 'café Über', ' en quad',
 'Hello World', 'HELLO WORLD', 'hello WORLD again', 'THE QUICK brown FOX', "DON'T STOP", "I'M OK", 'O’NEIL', "ROCK'N'ROLL", 'A-B', 'A.B.C.',
 'MiXeD cAsE', 'x1Y2z3', '3D', '2ND', 'ÉCOLE', 'École', 'ÜBER', 'İ', 'İstanbul', 'ǅ', 'STRASSE', 'Ⅷ', 'ΑΒΓ αβγ', 'ДА нет', '1st 2ND 3Rd',
 'Bϒa', "B'ϔ0Ba", 'BϔBa', 'ϒa', 'Bϒ', 'ϓΑβ',       // capitals without a lower-case form (round 5: found by tools/norm_masks_check.cpp, seed 21)
 ' W', 'C D W', 'DW', ' D', 'a  B', 'a\tB', 'a\nB', "'A", "'a", "1'a", "a'1", 'áB', 'Áb', 'ÁB', 'Á', 'ÁB', 'Áb',
].forEach(s => inputs.push(s));
for (let i = 0; i < 6300; i++) {
  const f = flavours[i % flavours.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

// round 4: the scripts the device normalizer / decoder took over from the host - Cyrillic and Greek (case, letters that decompose into a
// two-byte letter + mark: й ё Й ά ώ ΐ), Hebrew and Arabic with their points, Arabic-Indic digits, Chinese, kana (voiced kana decompose),
// Hangul - appended BEHIND the cases above so that those stay what they were
const cyr = 'абвгдежзийклмнопрстуфхцчшщъыьэюяёАБВГДЕЖЗИЙКЛМНОПРСТУФХЦЧШЩЪЫЬЭЮЯЁѐѝЍї'.split('');
const grk = 'αβγδεζηθικλμνξοπρστυφχψωςάέήίόύώϊϋΐΰΑΒΓΔΕΖΗΘΙΚΛΜΝΞΟΠΡΣΤΥΦΧΨΩΆΈΉΊΌΎΏ'.split('');
const heb = 'אבגדהוזחטיכלמנסעפצקרשתךםןףץ'.split('').concat(['ְ', 'ִ', 'ֵ', 'ָ', 'ּ']);
const arb = 'ابتثجحخدذرزسشصضطظعغفقكلمنهوي'.split('').concat(['َ', 'ُ', 'ِ', 'ّ', '٠', '١', '٢', '٣', '٩', 'آ', 'أ', 'ؤ']);
const cjk = '中文字符测试世界你好日本語漢字東京大学人工智能，。、「」！？（）'.split('');
const kana = 'あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんアイウエオカキクケコガギグゲゴパピプペポばびぶべぼーっッ'.split('');
const hangul = '한국어텍스트가나다'.split('');
const flavours4 = [
  () => pick([cyr, cyr, lower, upper, [' '], [' '], digits, apos, punct]),
  () => pick([grk, grk, lower, upper, [' '], [' '], digits, apos, marks]),
  () => pick([heb, arb, lower, upper, [' '], digits, punct]),
  () => pick([cjk, cjk, kana, lower, upper, [' '], digits, gpunct]),
  () => pick([cyr, grk, cjk, kana, hangul, heb, arb, lower, upper, [' '], apos, digits, marks]),
];
['Привет, Мир! Ёжик и йод. МОСКВА', 'Καλημέρα κόσμε. ΑΘΗΝΑ ά ώ ΐ Σίσυφος ΟΔΟΣ', 'שלום עולם בְּרֵאשִׁית', 'مرحبا بالعالم ١٢٣ كِتَاب', '中文文本，测试。Hello世界 ABC中文',
 'こんにちは世界 カタカナ がぎぐ パピプ', '한국어 텍스트', 'Йод йод ЙОД', "ДОН'Т д'Артаньян", 'x1Й2й3'].forEach(s => inputs.push(s));
for (let i = 0; i < 3000; i++) {
  const f = flavours4[i % flavours4.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

// round 5: what the device normalizer / decoder took over last - the characters beyond the Basic Multilingual Plane (four bytes of UTF-8:
// emoji and pictographs, their skin-tone modifiers, mathematical letters, plane-2 ideographs, Deseret with its case, a musical symbol that
// decomposes) and Hangul syllables, which NFD decomposes by arithmetic into two or three conjoining jamo - again appended BEHIND everything else
const emoji = ['😀', '😂', '🚀', '🌍', '👍', '🏽', '🎉', '🔥', '💯', '🤖', '🦄', '🧠', '🫠', '🇩', '🇪', '🀄', '🂡', '❤️', '☺️', '1️⃣', '\ufe0f', '✨'];
const astral = ['𝒜', '𝒷', '𝔘', '𝟘', '𠀀', '𠮷', '𪚥', '𐐀', '𐐨', '𝅗𝅥', '𐀀', '𒀀', '𓀀'];
const syll = '가각갂힣뷁한국어텍스트나다라마바사아자차카타파하값삶닭없읽'.split('');
const flavours5 = [
  () => pick([emoji, emoji, lower, upper, [' '], [' '], digits, apos, punct]),
  () => pick([emoji, astral, lower, upper, [' '], apos, digits, marks, gpunct]),
  () => pick([syll, syll, syll, [' '], [' '], lower, upper, digits, punct]),
  () => pick([syll, emoji, cjk, kana, cyr, accented, lower, upper, [' '], apos, digits]),
  () => pick([words, words, emoji, [' '], [' '], punct, syll]),
];
['Hello 😀 World 🌍🚀', "it's 👍🏽 A😀B c😀d 1😀2 '😀'", '😀', ' 😀', '😀A', 'A😀', '😀a', 'AB😀CD', 'Ab😀cD', '𝒜𝒷 𠀀𠮷 done', '𐐀𐐨 Deseret', '𝅗𝅥 half',
 '한국어 텍스트', '가', '각', '힣', 'A가B 가a 1가', "'가' 한글Hangul", '대한민국 KOREA 서울 Seoul', '값 삶 닭 없다 읽다', '가́', '🇩🇪 🇰🇷', 'I ❤️ U', 'A️b a️B ️', '1️⃣2️⃣ #️⃣', 'हिन्दी की कि HINDI'].forEach(s => inputs.push(s));
for (let i = 0; i < 2500; i++) {
  const f = flavours5[i % flavours5.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

// round 6: Latin Extended Additional (U+1E00..U+1EFF) - what Vietnamese is written in beside the two-byte letters: NFD makes a letter and one
// or two combining marks of each character, and the device normalizer now does that itself.  Appended BEHIND everything else once more.
const viet = 'ếệềểễấậầẩẫắặằẳẵớợờởỡứựừửữạảịỉọỏụủỳỵỷỹẾỆẤẬẮẶỚỢỨỰẠẢỊỌỤỲỸ'.split('');
const viet2 = 'đĐơƠưƯăĂâÂêÊôÔáàãéèíìóòõúùýÁÀÉÈ'.split('');
// (not ẞ U+1E9E: capcode writes its lower-case form ß, and what the decoder makes of a capitalised ß depends on the case mapping at hand -
// this JavaScript's toUpperCase gives "SS" (the full mapping), ICU's u_toupper, which the host decoder and the checker use, leaves ß alone (the
// simple one); which of the two Go's unicode.ToUpper is cannot be run here.  The ENCODER agrees on it everywhere; it is left out rather than pinned to one side.)
const lea = ['Ḁ', 'ḁ', 'ẛ', 'ỿ', 'ṩ', 'Ḉ', 'ḉ', 'ẘ', 'ẙ', 'ẚ'];
const vwords = ['Việt', 'Nam', 'tiếng', 'Hà', 'Nội', 'phố', 'Hồ', 'Chí', 'Minh', 'đường', 'Nguyễn', 'phở', 'ĐƯỜNG', 'TIẾNG', 'VIỆT', 'người', 'được', 'ƯỚC', 'Ắt', 'the', 'HTTP', 'it', 's'];
const flavours6 = [
  () => pick([viet, viet, viet2, lower, upper, [' '], [' '], digits, apos, punct]),
  () => pick([vwords, vwords, [' '], [' '], apos, punct, digits, upper]),
  () => pick([viet, lea, marks, lower, upper, [' '], apos, digits, gpunct]),
  () => pick([viet, viet2, accented, cyr, cjk, syll, emoji, lower, upper, [' '], apos, digits]),
];
['Việt Nam: tiếng Việt, Hà Nội và Thành phố Hồ Chí Minh.', 'ĐƯỜNG Nguyễn Huệ, PHỞ bò', 'TIẾNG VIỆT viết HOA và Thường', "ớt's Ớt'S 1ế2 'ệ'", 'Ắ ắ Ế ế Ộ ộ Ự ự Ỹ ỹ', 'Ḁḁ ẛ ỿ', 'ế́ ệ̣ ế',
 'ế', 'Ế', ' Ế', 'ẾỆ', 'Ếệ', 'aẾ', 'ẾA', 'Ế1', '1Ế', "Ế'S", 'ẾẾẾ x ẾẾb'].forEach(s => inputs.push(s));
for (let i = 0; i < 2000; i++) {
  const f = flavours6[i % flavours6.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

// round 6, second half: the voiced kana (が -> か + U+3099, ぱ -> は + U+309A, ヴ ヷ ヸ ヹ ヺ ゞ ヾ under NFD) - what Japanese running text is full of;
// the device normalizer now decomposes them itself.  Appended BEHIND everything else again.
const kana7 = 'あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんアイウエオカキクケコサシスセソタチツテトハヒフヘホ'.split('');
const voiced = 'がぎぐげござじずぜぞだぢづでどばびぶべぼぱぴぷぺぽゔゞガギグゲゴザジズゼゾダヂヅデドバビブベボパピプペポヴヷヸヹヺヾ'.split('');
const jwords = ['です', 'ございます', 'がんばって', 'データ', 'プログラム', 'ヴァイオリン', '日本語', '東京', 'は', 'の', '、', '。', 'Tokyo', 'GPU', 'it', 's', 'ABC'];
const flavours7 = [
  () => pick([kana7, voiced, voiced, cjk, lower, upper, [' '], digits, apos]),
  () => pick([jwords, jwords, [''], [' '], apos, punct, digits, upper]),
  () => pick([voiced, ['\u3099', '\u309A'], marks, kana7, lower, upper, [" "], apos, digits]),
  () => pick([voiced, kana7, viet, accented, cyr, syll, emoji, lower, upper, [' '], apos, digits]),
];
['がぎぐげご ひらがなの濁点', 'パピプペポ と ばびぶべぼ', 'ヴァイオリンのデータです。', "Aが Bガ'S 1ぱ2 'ぴ'", 'がA ガb GAが', 'か\u3099 は\u309A が\u0301 が\u3099', 'ゞ ヾ ゔ ヷ ヸ ヹ ヺ',
 'が', 'ガ', ' が', 'がが', 'aが', 'がa', 'が1', "が'S", 'TOKYOでGPUをつかう'].forEach(s => inputs.push(s));
for (let i = 0; i < 1200; i++) {
  const f = flavours7[i % flavours7.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

// round 6, third part: the three-byte combining marks of canonical class > 0 (virama, nukta, the Thai tone marks and vowels below ...) and the
// three-byte digits - Hindi and Thai running text, marks in and out of canonical order.  Appended BEHIND everything else again.
const deva = 'कखगघचछजझटठडढणतथदधनपफबभमयरलवशषसहअआइईउऊएऐओऔ'.split('');
const devam = ['\u093e', '\u093f', '\u0940', '\u0941', '\u0942', '\u0947', '\u0948', '\u094b', '\u094c', '\u094d', '\u094d', '\u093c', '\u0902', '\u0903', '\u0901', '\u0951', '\u0952'];
const devad = '०१२३४५६७८९'.split('');
const thai = 'กขคงจฉชซญดตถทธนบปผฝพฟภมยรลวศษสหอฮะาำเแโใไ'.split('');
const thaim = ['\u0e31', '\u0e34', '\u0e35', '\u0e36', '\u0e37', '\u0e38', '\u0e39', '\u0e3a', '\u0e47', '\u0e48', '\u0e49', '\u0e4a', '\u0e4b', '\u0e4c'];
const thaid = '๐๑๒๓๔๕๖๗๘๙'.split('');
const hwords = ['यह', 'हिन्दी', 'का', 'पाठ', 'है', 'विश्वविद्यालय', 'प्रौद्योगिकी', 'स्वतंत्रता', 'ज\u093cिन्दगी', 'फ\u093cिल्म', 'क्या', 'भारत', 'दिल्ली', 'ภาษาไทย', 'อยู่', 'ที่', 'กรุงเทพมหานคร', 'น้ำ', 'ผู้', 'ใหญ่', 'รู้', 'เรื่อง', 'GPU', 'it', 's', 'ABC', '१२३', '๑๒๓'];
const flavours8 = [
  () => pick([deva, deva, devam, devam, devad, lower, upper, [' '], digits, apos]),
  () => pick([thai, thai, thaim, thaim, thaid, lower, upper, [' '], digits, apos]),
  () => pick([hwords, hwords, [''], [' '], apos, punct, digits, upper]),
  () => pick([deva, thai, devam, thaim, marks, ['\u0929', '\u0958', '\u09cb', '\u0bca', '\u0f73', '\u1026'], lower, upper, [' '], apos, digits, devad]),
];
['यह हिन्दी का पाठ है।', 'विश्वविद्यालय 2024 में GPU', 'ภาษาไทย อยู่ที่กรุงเทพมหานคร', 'น้ำ ผู้ใหญ่ รู้เรื่อง', "Aक Bก'S 1क्2 'ก่'", 'कA กb १a a१ ๑A', 'क\u094d\u093c क\u093c\u094d', 'ก\u0e48\u0e38 ก\u0e38\u0e48',
 'क\u0301 ก\u0301 \u1e09\u0e48 é\u094d', '१२३ ๑๒๓ 1१ १1', 'ໄທ ລາວ ພາສາ', 'བོད་སྐད', 'မြန်မာ', 'ខ្មែរ', 'தமிழ் ಕನ್ನಡ తెలుగు മലയാളം ગુજરાતી ਪੰਜਾਬੀ বাংলা'].forEach(s => inputs.push(s));
for (let i = 0; i < 1600; i++) {
  const f = flavours8[i % flavours8.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

// round 6, fourth part: lower-case letters of three bytes (Georgian Mkhedruli, the phonetic extensions) - class L to capcode, nothing changes them;
// their capitals (Mtavruli) stay the host's.  Appended BEHIND everything else again.
const geo = 'აბგდევზთიკლმნოპჟრსტუფქღყშჩცძწჭხჯჰ'.split('');
const gwords = ['საქართველო', 'თბილისი', 'ქართული', 'ენა', 'და', 'არის', 'ᴀ', 'ᴛᴇxᴛ', 'ᲡᲐᲥᲐᲠᲗᲕᲔᲚᲝ', 'Tbilisi', 'GPU', 'it', 's', '2024'];
const flavours9 = [
  () => pick([geo, geo, lower, upper, [' '], digits, apos, punct]),
  () => pick([gwords, gwords, [' '], [' '], apos, punct, digits, upper]),
  () => pick([geo, ['ᴀ', 'ᴇ', 'ᴛ', 'ᵃ', 'ᶜ', 'Ა', 'Ბ', 'Ꭰ', 'ꭰ'], marks, lower, upper, [' '], apos, digits]),
];
['საქართველო არის ქვეყანა.', "Aა Bბ'S 1გ2 'დ'", 'აA აb GAა ააა', 'ქართული ენა 2024 GPU', 'ᴛᴇxᴛ in small caps', 'ა\u0301 ა́ბ', 'ᲡᲐᲥᲐᲠᲗᲕᲔᲚᲝ Mtavruli'].forEach(s => inputs.push(s));
for (let i = 0; i < 900; i++) {
  const f = flavours9[i % flavours9.length];
  const n = rnd(rnd(4) === 0 ? 90 : 28);
  let s = '';
  for (let k = 0; k < n; k++) s += pick(f());
  inputs.push(s);
}

const b64 = s => Buffer.from(s, 'utf8').toString('base64');
const cases = inputs.map(s => {
  const nfd = s.normalize('NFD');
  const enc = ctx.capcode_encode(nfd);
  const dec = new ctx.CapcodeDecoder().decode(enc);
  return { in: b64(s), nfd: b64(nfd), enc: b64(enc), dec: b64(dec) };
});
process.stdout.write(JSON.stringify({
  source: 'javascript/tokenmonster.js:872-1065 evaluated in place by node ' + process.version + ' (tests/golden/make_capcode_golden.js)',
  unicode: process.versions.unicode, icu: process.versions.icu, n: cases.length, cases,
}));

"""tokenmonster_amd — MI355X-native batch tokenizer for TokenMonster vocabularies.

Host-side mirror of the reference's tokenize entry points (go/tokenmonster.go:953-1014,
python/tokenmonster.py:410 `Vocab.tokenize`, :497 `tokenize_count`) above the C ABI of
libtokenmonster_hip.so.  All tokenization runs in hand-written HIP kernels on gfx950."""
from .vocab import Decoder, PinnedBuffer, Vocab, VocabBlock, load, pack_documents  # noqa: F401
from . import synth  # noqa: F401

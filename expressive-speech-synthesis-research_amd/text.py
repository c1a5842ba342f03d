"""Phoneme-level tokenizer with the reference's vocabulary.

Same table and index assignment as TransformerTTS/preprocessing/text/symbols.py:1-12 and
tokenizer.py:4-27 (index 0 = padding, symbols sorted by code point from 1, optional start/end
tokens appended), so token ids interchange with data prepared by the reference.  The espeak
phonemizer is an external program and is not part of this package: callers pass phoneme strings.
"""

_VOWELS = 'iyɨʉɯuɪʏʊeøɘəɵɤoɛœɜɞʌɔæɐaɶɑɒᵻ'
_NON_PULMONIC = 'ʘɓǀɗǃʄǂɠǁʛ'
_PULMONIC = 'pbtdʈɖcɟkɡqɢʔɴŋɲɳnɱmʙrʀⱱɾɽɸβfvθðszʃʒʂʐçʝxɣχʁħʕhɦɬɮʋɹɻjɰlɭʎʟ'
_SUPRASEGMENTALS = 'ˈˌːˑ'
_OTHER = 'ʍwɥʜʢʡɕʑɺɧ'
_DIACRITICS = 'ɚ˞ɫ'
_PUNCTUATION = '!,-.:;? '


def default_alphabet():
    phonemes = sorted(list(_VOWELS + _NON_PULMONIC + _PULMONIC + _SUPRASEGMENTALS + _OTHER + _DIACRITICS))
    return sorted(phonemes + list(_PUNCTUATION))


class Tokenizer:
    def __init__(self, alphabet, start_token='>', end_token='<', pad_token='/', add_start_end=True):
        self.alphabet = list(alphabet)
        self.idx_to_token = {0: pad_token}
        for i, s in enumerate(self.alphabet, start=1):
            self.idx_to_token[i] = s
        self.token_to_idx = {s: i for i, s in self.idx_to_token.items()}
        self.vocab_size = len(self.alphabet) + 1
        self.add_start_end = add_start_end
        if add_start_end:
            self.start_token_index = self.vocab_size
            self.end_token_index = self.vocab_size + 1
            self.idx_to_token[self.start_token_index] = start_token
            self.idx_to_token[self.end_token_index] = end_token
            self.vocab_size += 2

    def __call__(self, sentence):
        ids = [self.token_to_idx[c] for c in sentence if c in self.token_to_idx]
        if self.add_start_end:
            ids = [self.start_token_index] + ids + [self.end_token_index]
        return ids

    def decode(self, sequence):
        return ''.join(self.idx_to_token[int(t)] for t in sequence if int(t) in self.idx_to_token)

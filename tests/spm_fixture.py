"""A REAL SentencePiece tokenizer for the tests: a unigram model trained on a synthetic corpus (there are no checkpoint files
offline), saved in the Hugging Face T5 layout (spiece.model + tokenizer_config.json) so that `AutoTokenizer.from_pretrained`
-- the call generate.py / main.py make -- loads it.  pad = 0, eos = 1, unk = 2 like t5-base."""
import json
import os
import random


def build_t5_tokenizer_dir(path, vocab_size=400):
    import sentencepiece as spm

    os.makedirs(path, exist_ok=True)
    rnd = random.Random(0)
    words = [f"w{i}" for i in range(300)] + "what is the capital of how many why does a an in on q".split() + [f"q{i}" for i in range(30)]
    corpus = os.path.join(path, "corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(3000):
            f.write(" ".join(rnd.choice(words) for _ in range(rnd.randint(3, 12))) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(path, "spiece"), vocab_size=vocab_size,
                                   model_type="unigram", pad_id=0, eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False,
                                   minloglevel=2)
    os.remove(corpus)
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "T5Tokenizer", "extra_ids": 0}, f)
    return path

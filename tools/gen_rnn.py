#!/usr/bin/env python3
"""Seeded synthetic faster-rnnlm model in the on-disk format the reference reads
(src/rnn/mikolov_rnn.cc:16-76 header, :163-215 payload order
 emb[V x E], nce[V x E], W[E x E], maxent[M]; version 6, NCE, sigmoid).
Vocabulary words are `<field1>_<field2>` strings (default surface_pos) sampled
from a dictionary CSV plus a few out-of-dictionary katakana words.

usage: gen_rnn.py <dict.mdic> <out-prefix> [--vocab 5000] [--hidden 128]
                  [--maxent-order 3] [--maxent-size 1048576] [--seed 1]
writes <out-prefix> (vocabulary) and <out-prefix>.nnet (weights)."""
import argparse
import random
import struct

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dict')
    ap.add_argument('out')
    ap.add_argument('--vocab', type=int, default=5000)
    ap.add_argument('--hidden', type=int, default=128)
    ap.add_argument('--maxent-order', type=int, default=3)
    ap.add_argument('--maxent-size', type=int, default=1 << 20)
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    pairs = []
    seen = set()
    with open(a.dict, encoding='utf-8') as f:
        for i, line in enumerate(f):
            if i < 9:
                continue
            p = line.rstrip('\n').split(',')
            if len(p) < 6 or '_' in p[0] or ' ' in p[0]:
                continue
            key = (p[0], p[4])
            if key not in seen:
                seen.add(key)
                pairs.append(key)
    rng.shuffle(pairs)
    words = ['</s>', '<unk>']
    words += ['%s_%s' % k for k in pairs[:a.vocab]]
    kata = 'アイウエオカキクケコサシスセソタチツテトナニヌネノ'
    for _ in range(max(20, a.vocab // 25)):
        w = ''.join(rng.choice(kata) for _ in range(rng.randint(2, 5)))
        words.append('%s_未定義語' % w)
    words = list(dict.fromkeys(words))
    with open(a.out, 'w', encoding='utf-8') as f:
        for i, w in enumerate(words):
            f.write('%s %d\n' % (w, len(words) - i))
    V, E, M = len(words), a.hidden, a.maxent_size
    nrng = np.random.RandomState(a.seed)
    with open(a.out + '.nnet', 'wb') as f:
        f.write(struct.pack('<Q', 6 * 10000 + E))
        f.write(struct.pack('<Q', M))
        f.write(struct.pack('<I', a.maxent_order))
        f.write(struct.pack('<B', 1))            # use_nce
        f.write(struct.pack('<f', 9.0))          # nce_lnz
        f.write(struct.pack('<B', 0))            # reversed sentence
        f.write(b'sigmoid'.ljust(64, b'\0'))     # layer type
        f.write(struct.pack('<I', 1))            # layer count
        f.write(struct.pack('<I', 0))            # hs arity
        for shape in ((V, E), (V, E), (E, E)):
            f.write(nrng.uniform(-0.3, 0.3, size=shape).astype('<f4').tobytes())
        f.write(nrng.uniform(-0.5, 0.5, size=(M,)).astype('<f4').tobytes())
    print('wrote %s: V=%d E=%d maxent order %d size %d' % (a.out, V, E, a.maxent_order, M))


if __name__ == '__main__':
    main()

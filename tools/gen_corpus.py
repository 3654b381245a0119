#!/usr/bin/env python3
"""Seeded synthetic corpus (SURVEY section 8(d) "Synthetic inputs"):
lines of exactly `--len` codepoints built from dictionary surfaces plus a share
of out-of-dictionary katakana / kanji / digit / ASCII runs and occasional
prolong marks, small kana and sokuon so every UNK maker (incl. the normalizer)
fires.

usage: gen_corpus.py <dict.mdic> <n_lines> [--len 40] [--seed 1] [--oov 0.05] [--zipf S] > corpus.txt

--zipf S: dictionary words are drawn with probability proportional to 1 / rank^S (rank over a seeded shuffle of the
surfaces) instead of uniformly -- the word statistics of real text, where a few thousand words make most of the tokens.
"""
import argparse
import random
import sys


def load_surfaces(path, limit=200000):
    out = []
    with open(path, encoding='utf-8') as f:
        for i, line in enumerate(f):
            if i < 9:
                continue  # UNK templates
            s = line.split(',', 1)[0]
            if s and not s.startswith('"') and '#' not in s[:1]:
                out.append(s)
            if len(out) >= limit:
                break
    return out


KATA = 'アイウエオカキクケコサシスセソタチツテトナニヌネノハヒフヘホマミムメモヤユヨラリルレロワヲンガギグゲゴザジズゼゾダヂヅデドバビブベボパピプペポァィゥェォッャュョヮヵヶ'
KANJI = '日本語形態素解析京都大学研究室東西南北山川田中村上下左右年月火水木金土兵数何幾百千万億兆一二三四五六七八九十零'
DIGITS = '0123456789０１２３４５６７８９'
ASCII = 'abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ'
HIRA = 'あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんがぎぐげござじずぜぞだでどばびぶべぼぱぴぷぺぽぁぃぅぇぉっゃゅょ'
SPECIAL = ['ー', '〜', 'っ', 'ッ', 'ぁ', 'ぃ', 'ぅ', 'ぇ', 'ぉ', '、', '。', '！', '？', '・', '，', '．', ',', '.',
           '（', '）', '「', '」', '％', 'キロ', 'メガ', 'ミリ', '数', '何', '分の', 'ぶんの', ' ', '　']


def make_line(rng, surfaces, length, oov, cum=None):
    parts = []
    n = 0
    while n < length:
        r = rng.random()
        if r < oov:
            kind = rng.randrange(7)
            k = rng.randint(1, 6)
            if kind == 0:
                w = ''.join(rng.choice(KATA) for _ in range(k))
            elif kind == 1:
                w = ''.join(rng.choice(KANJI) for _ in range(rng.randint(1, 3)))
            elif kind == 2:
                w = ''.join(rng.choice(DIGITS) for _ in range(k))
                if rng.random() < 0.3:
                    w += rng.choice([',', '，', '.', '．', '・']) + ''.join(rng.choice(DIGITS) for _ in range(3))
            elif kind == 3:
                w = ''.join(rng.choice(ASCII) for _ in range(k))
            elif kind == 4:
                w = rng.choice(SPECIAL)
            elif kind == 5:
                h = ''.join(rng.choice(KATA + HIRA) for _ in range(rng.randint(2, 4)))
                w = h + h  # onomatopoeia shape
            else:
                w = ''.join(rng.choice(HIRA) for _ in range(k))
                if rng.random() < 0.5:
                    w += rng.choice(['ー', '〜', 'っ', 'ぁ', 'ぇ', 'ーー'])
        else:
            w = rng.choice(surfaces) if cum is None else rng.choices(surfaces, cum_weights=cum)[0]
        parts.append(w)
        n += len(w)
    line = ''.join(parts)[:length]
    line = line.replace('\n', '').replace('\r', '')
    while line.startswith('# '):
        line = 'あ' + line[1:]
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dict')
    ap.add_argument('n', type=int)
    ap.add_argument('--len', type=int, default=40)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--oov', type=float, default=0.05)
    ap.add_argument('--zipf', type=float, default=0.0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    surfaces = load_surfaces(a.dict)
    cum = None
    if a.zipf > 0:
        random.Random(a.seed ^ 0x5bd1e995).shuffle(surfaces)
        cum, acc = [], 0.0
        for r in range(len(surfaces)):
            acc += 1.0 / float(r + 1) ** a.zipf
            cum.append(acc)
    out = sys.stdout
    for _ in range(a.n):
        out.write(make_line(rng, surfaces, a.len, a.oov, cum) + '\n')


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Seeded synthetic dictionary in the jumandic CSV column layout
(surface,0,0,0,pos,subpos,conjform,conjtype,baseform,reading,canonic,features;
reference src/jumandic/shared/jumandic_spec.cc:31-43).  No reference data is
used: surfaces are random kana/kanji strings with a Zipf-like length/letter
distribution, part-of-speech columns are sampled from a hand-written table of
JUMAN tag combinations.  Rows 1-8 are the UNK templates the jumandic spec
points at (jumandic_spec.cc:67-100).

usage: gen_dict.py <n_entries> [--seed 1] [--homographs K] > dict.mdic

--homographs K: jumandic-like fan-out of the short, frequent surfaces -- every single hiragana gets 8..K entries
(particles, verb stems, suffixes, ... of one surface), 150 two-kana surfaces get 4..K/2 -- so that boundaries with
dozens of right nodes occur in ordinary sentences (the rest of the dictionary keeps at most 6 entries per surface).
"""
import argparse
import random
import sys

UNK_ROWS = [
    'UNK_SYM,0,0,0,未定義語,その他,*,*,UNK,UNK,*,品詞推定:特殊 未知語:その他',
    'UNK_KATA,0,0,0,未定義語,カタカナ,*,*,UNK,UNK,*,品詞推定:名詞 未知語:カタカナ',
    'UNK_KANJI,0,0,0,未定義語,その他,*,*,UNK,UNK,*,品詞推定:名詞 未知語:漢字',
    'UNK_HIRA,0,0,0,未定義語,その他,*,*,UNK,UNK,*,品詞推定:名詞 未知語:ひらがな',
    'UNK_ALPH,0,0,0,未定義語,アルファベット,*,*,UNK,UNK,*,品詞推定:名詞 未知語:ロマ字',
    'UNK_DIGIT,0,0,0,名詞,数詞,*,*,UNK,UNK,*,カテゴリ:数量 未知語:数字',
    'UNK_ONOMATOPEA,0,0,0,副詞,*,*,*,UNK,UNK,*,自動認識 未知語:オノマトペ',
    'UNK_ANY,0,0,0,未定義語,その他,*,*,UNK,UNK,*,品詞推定:特殊 未知語:未対応文字種',
]

# (pos, subpos, [(conjform, conjtype, okurigana)], weight)
TAGS = [
    ('名詞', '普通名詞', [('*', '*', '')], 30),
    ('名詞', 'サ変名詞', [('*', '*', '')], 8),
    ('名詞', '固有名詞', [('*', '*', '')], 4),
    ('名詞', '地名', [('*', '*', '')], 4),
    ('名詞', '人名', [('*', '*', '')], 4),
    ('名詞', '組織名', [('*', '*', '')], 2),
    ('名詞', '時相名詞', [('*', '*', '')], 1),
    ('名詞', '形式名詞', [('*', '*', '')], 1),
    ('名詞', '副詞的名詞', [('*', '*', '')], 1),
    ('動詞', '*', [('基本形', '母音動詞', 'る'), ('未然形', '母音動詞', ''), ('基本連用形', '母音動詞', ''),
                    ('タ形', '母音動詞', 'た'), ('タ系連用テ形', '母音動詞', 'て')], 10),
    ('動詞', '*', [('基本形', '子音動詞カ行', 'く'), ('未然形', '子音動詞カ行', 'か'), ('基本連用形', '子音動詞カ行', 'き'),
                    ('タ形', '子音動詞カ行', 'いた')], 6),
    ('動詞', '*', [('基本形', '子音動詞ラ行', 'る'), ('未然形', '子音動詞ラ行', 'ら'), ('基本連用形', '子音動詞ラ行', 'り'),
                    ('タ形', '子音動詞ラ行', 'った')], 6),
    ('形容詞', '*', [('基本形', 'イ形容詞アウオ段', 'い'), ('語幹', 'イ形容詞アウオ段', ''),
                      ('基本連用形', 'イ形容詞アウオ段', 'く'), ('タ形', 'イ形容詞アウオ段', 'かった')], 5),
    ('形容詞', '*', [('基本形', 'ナ形容詞', 'だ'), ('語幹', 'ナ形容詞', ''), ('ダ列基本連体形', 'ナ形容詞', 'な')], 4),
    ('副詞', '*', [('*', '*', '')], 4),
    ('連体詞', '*', [('*', '*', '')], 1),
    ('接続詞', '*', [('*', '*', '')], 1),
    ('感動詞', '*', [('*', '*', '')], 1),
    ('助詞', '格助詞', [('*', '*', '')], 1),
    ('助詞', '副助詞', [('*', '*', '')], 1),
    ('助詞', '接続助詞', [('*', '*', '')], 1),
    ('助詞', '終助詞', [('*', '*', '')], 1),
    ('助動詞', '*', [('基本形', 'ナ形容詞', 'だ'), ('語幹', 'ナ形容詞', '')], 1),
    ('判定詞', '*', [('基本形', '判定詞', 'だ'), ('デアル列基本形', '判定詞', 'である')], 1),
    ('接尾辞', '名詞性名詞接尾辞', [('*', '*', '')], 2),
    ('接尾辞', '名詞性名詞助数辞', [('*', '*', '')], 1),
    ('接頭辞', '名詞接頭辞', [('*', '*', '')], 1),
    ('指示詞', '名詞形態指示詞', [('*', '*', '')], 1),
    ('特殊', '句点', [('*', '*', '')], 0),
    ('特殊', '読点', [('*', '*', '')], 0),
]

HIRA = 'あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをんがぎぐげござじずぜぞだでどばびぶべぼぱぴぷぺぽ'
KATA = 'アイウエオカキクケコサシスセソタチツテトナニヌネノハヒフヘホマミムメモヤユヨラリルレロワンガギグゲゴザジズゼゾダデドバビブベボパピプペポー'
FEATURES = ['NIL', 'NIL', 'NIL', '"代表表記:x/x"', '"カテゴリ:抽象物"', '"濁音化D"', '"連用形名詞化"', '"カテゴリ:人 ドメイン:政治"']


def kanji_pool(n):
    # CJK unified ideographs, deterministic slice
    return [chr(0x4E00 + 7 * i % 0x5000) for i in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('n', type=int)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--homographs', type=int, default=0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    kanji = kanji_pool(3000)
    kw = [1.0 / (i + 20) ** 0.5 for i in range(len(kanji))]
    per_surface = {}
    MAX_HOMOGRAPHS = 6
    out = sys.stdout
    for r in UNK_ROWS:
        out.write(r + '\n')
    fixed = [('、', '特殊', '読点'), ('。', '特殊', '句点'), ('の', '助詞', '接続助詞'), ('は', '助詞', '副助詞'),
             ('が', '助詞', '格助詞'), ('を', '助詞', '格助詞'), ('に', '助詞', '格助詞'), ('で', '助詞', '格助詞'),
             ('と', '助詞', '格助詞'), ('も', '助詞', '副助詞'), ('だ', '判定詞', '*'), ('た', '助動詞', '*')]
    for s, p, sp in fixed:
        cf = '基本形' if p in ('判定詞', '助動詞') else '*'
        ct = '判定詞' if p == '判定詞' else ('ナ形容詞' if p == '助動詞' else '*')
        out.write('%s,0,0,0,%s,%s,%s,%s,%s,%s,*,NIL\n' % (s, p, sp, cf, ct, s, s))
    weights = [t[3] for t in TAGS]
    seen = set()
    count = len(fixed)
    if a.homographs > 0:
        hrng = random.Random(a.seed * 7919 + 13)
        shorts = [(c, hrng.randint(8, max(8, a.homographs))) for c in HIRA]
        two = set()
        while len(two) < 150:
            two.add(hrng.choice(HIRA) + hrng.choice(HIRA))
        shorts += [(w, hrng.randint(4, max(4, a.homographs // 2))) for w in sorted(two)]
        plain = [t for t in TAGS if t[2] == [('*', '*', '')] and t[3] > 0]
        for surf, k in shorts:
            for j in range(k):
                pos, subpos, _, _ = plain[(j * 7 + len(surf)) % len(plain)]
                rd = ''.join(hrng.choice(HIRA) for _ in range(len(surf)))
                out.write('%s,0,0,0,%s,%s,*,*,%s,%s,%s,%s\n' % (surf, pos, subpos, surf, rd, surf + '/' + rd,
                                                                  hrng.choice(FEATURES)))
                count += 1
            per_surface[surf] = 1 << 30   # no further entries from the generic generator
    while count < a.n:
        pos, subpos, forms, _ = rng.choices(TAGS, weights=weights)[0]
        style = rng.random()
        ln = min(1 + int(rng.expovariate(0.9)), 6)
        if style < 0.6:
            stem = ''.join(rng.choices(kanji, weights=kw, k=max(1, min(ln + (1 if rng.random() < 0.6 else 0), 4))))
            if rng.random() < 0.3:
                stem += ''.join(rng.choice(HIRA) for _ in range(rng.randint(1, 2)))
        elif style < 0.8:
            stem = ''.join(rng.choice(HIRA) for _ in range(ln + 1))
        else:
            stem = ''.join(rng.choice(KATA) for _ in range(ln + 1))
        reading_stem = ''.join(rng.choice(HIRA) for _ in range(max(1, len(stem))))
        base = stem + forms[0][2]
        feat = rng.choice(FEATURES)
        nread = 2 if rng.random() < 0.08 else 1  # homographs with different readings -> aliased/tied entries
        for rd in range(nread):
            rstem = reading_stem if rd == 0 else ''.join(rng.choice(HIRA) for _ in range(len(reading_stem)))
            for cf, ct, oku in forms:
                surf = stem + oku
                key = (surf, pos, subpos, cf, ct, rstem)
                if key in seen or not surf:
                    continue
                if per_surface.get(surf, 0) >= MAX_HOMOGRAPHS:
                    continue
                per_surface[surf] = per_surface.get(surf, 0) + 1
                seen.add(key)
                out.write('%s,0,0,0,%s,%s,%s,%s,%s,%s,%s,%s\n' % (
                    surf, pos, subpos, cf, ct, base, rstem + oku, '*' if rng.random() < 0.5 else base + '/' + rstem,
                    feat))
                count += 1


if __name__ == '__main__':
    main()

"""The two inequalities the search over sets with more than four symbols rests on (csrc/nn_images.inc nn_phase_a_images):
with f = the class-merging map of the planes (lower case onto upper case, every byte outside ACGT onto one code) and
e(s) = number of bytes of s outside ACGT,
        d(f(x), f(y))  <=  d(x, y)  <=  d(f(x), f(y)) + e(x) + e(y).
Checked with the textbook DP of the oracle on random strings (no GPU)."""
import random

from oracle import oracle as O


def image(s):
    return "".join(c if c in "ACGT" else (c.upper() if c.upper() in "ACGT" else "T") for c in s)


def outside(s):
    return sum(1 for c in s if c not in "ACGT")


def test_image_distance_brackets_the_distance():
    rng = random.Random(5)
    tight_low = tight_high = 0
    for case in range(400):
        alphabet = rng.choice(["ACGTN", "ACGTacgt", "ACGTNnRY-", "ACGTacgtN"])
        weights = [8 if c in "ACGT" else 1 for c in alphabet]
        x = "".join(rng.choices(alphabet, weights, k=rng.randrange(0, 120)))
        y = list(x)
        for _ in range(rng.randrange(0, 25)):
            r = rng.random()
            if y and r < 0.4:
                y[rng.randrange(len(y))] = rng.choices(alphabet, weights)[0]
            elif y and r < 0.7:
                del y[rng.randrange(len(y))]
            else:
                y.insert(rng.randrange(len(y) + 1), rng.choices(alphabet, weights)[0])
        y = "".join(y)
        d, di = O.ed_dp(x, y), O.ed_dp(image(x), image(y))
        assert di <= d <= di + outside(x) + outside(y), (x, y, d, di)
        tight_low += di == d
        tight_high += di < d
    assert tight_low > 50 and tight_high > 50          # both regimes occur: images often keep the distance, and often lose some of it

#!/usr/bin/env python3
"""The closed form of the reference's in-place partition loop (bvh.cpp:82-93) that ct_build.hip uses, checked
against the loop itself on random inputs."""
import random


def loop(a, is_left):
    a, i, j = list(a), 0, len(a) - 1
    while i <= j:
        if is_left(a[i]):
            i += 1
        else:
            a[i], a[j] = a[j], a[i]
            j -= 1
    return a, i


def closed_form(a, is_left):
    n = len(a)
    right = [p for p in range(n) if not is_left(a[p])]              # r_1 < r_2 < ...
    left = [p for p in range(n - 1, -1, -1) if is_left(a[p])]       # l_1 > l_2 > ...
    pairs = 0
    while pairs < min(len(right), len(left)) and right[pairs] < left[pairs]:
        pairs += 1
    out = [None] * n
    for p in range(n):
        if is_left(a[p]) and p not in left[:pairs]:
            out[p] = a[p]
    for m in range(pairs):
        out[right[m]] = a[left[m]]
    last_paired_left = n if pairs == 0 else left[pairs - 1]
    for m, p in enumerate(right):
        if m < pairs or (m == pairs and p < last_paired_left):
            dst = (n if m == 0 else left[m - 1]) - 1
        else:
            dst = p - 1
        out[dst] = a[p]
    return out, len(left)


if __name__ == "__main__":
    random.seed(1)
    for trial in range(300000):
        n = random.randint(1, 14)
        density = random.choice([0.1, 0.3, 0.5, 0.7, 0.9])
        a = [(random.random() < density, k) for k in range(n)]
        assert loop(a, lambda e: e[0]) == closed_form(a, lambda e: e[0]), a
    print("closed form == swap loop on 300000 random arrays")

#!/usr/bin/env python3
"""Extract the reference's golden vectors (DATA only) into JSON fixtures.

Runs only in the build container, where /root/reference exists; the GPU box
and the test-suite read the committed JSON files, never the reference.

Sources (inputs + expected outputs, no code):
  * fft/_test_values.mojo:8-1106   -- 1-D (series, spectrum) pairs, N in
    {2,3,4,5,6,7,8,10,16,20,21,30,32,35,48,60,64,100,128}
  * fft/tests.mojo:422-458         -- input_2d (6x4 uint8), expected_2d
  * fft/tests.mojo:613-905         -- input_3d (6x4x8 uint8), expected_3d
  * fft/tests.mojo:280-371         -- the (N, bases) matrix the 1-D tests run

Usage: python tests/golden/extract_golden.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def _complex_list(text, ctor):
    out = []
    for m in re.finditer(ctor + r"\(\s*(" + NUM + r")\s*(?:,\s*(" + NUM + r")\s*)?\)", text):
        out.append([float(m.group(1)), float(m.group(2)) if m.group(2) else 0.0])
    return out


def extract_1d():
    src = open(os.path.join(REF, "fft", "_test_values.mojo")).read()
    res = {}
    # split per function
    parts = re.split(r"def _get_test_values_(\d+)\[", src)
    for k in range(1, len(parts), 2):
        n = int(parts[k])
        body = parts[k + 1]
        body = body[body.index("res = [") + len("res = ["):]
        pairs = []
        # each test case is "{ [ints], [Complex(...), ...] }"
        for m in re.finditer(r"\{\s*\[([^\]]*)\]\s*,\s*\[(.*?)\]\s*,?\s*\}", body, re.S):
            series = [int(v) for v in re.findall(r"[-+]?\d+", m.group(1))]
            spectrum = _complex_list(m.group(2), "Complex")
            assert len(series) == n and len(spectrum) == n, (n, len(series), len(spectrum))
            pairs.append({"x": series, "X": spectrum})
        assert pairs, n
        res[str(n)] = pairs
    return res


def extract_matrix():
    src = open(os.path.join(REF, "fft", "tests.mojo")).read()
    body = src[src.index("def _test_fft["):src.index("comptime _test[")]
    matrix = []
    for line in body.splitlines():
        line = line.strip()
        enabled = not line.startswith("#")
        m = re.search(r"func\[\[([\d,\s]+)\],\s*values_(\d+)\]\(\)", line)
        if m:
            matrix.append({
                "n": int(m.group(2)),
                "bases": [int(v) for v in m.group(1).split(",")],
                "enabled_in_reference": enabled,
            })
    return matrix


def _nested_ints(text):
    return json.loads(re.sub(r",\s*\]", "]", text))


def extract_nd():
    src = open(os.path.join(REF, "fft", "tests.mojo")).read()

    def block(name):
        i = src.index("comptime " + name)
        i = src.index("= [", i) + 2
        depth, j = 0, i
        while True:
            if src[j] == "[":
                depth += 1
            elif src[j] == "]":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        return src[i:j + 1]

    in2 = _nested_ints(block("input_2d"))
    ex2 = _complex_list(block("expected_2d"), "Co")
    assert len(in2) == 6 and len(in2[0]) == 4 and len(ex2) == 24
    in3 = _nested_ints(block("input_3d"))
    ex3 = _complex_list(block("expected_3d"), "Co")
    assert len(in3) == 6 and len(in3[0]) == 4 and len(in3[0][0]) == 8 and len(ex3) == 192
    return (
        {"shape": [6, 4], "in_dtype": "uint8", "x": in2, "X_flat": ex2},
        {"shape": [6, 4, 8], "in_dtype": "uint8", "x": in3, "X_flat": ex3},
    )


def main():
    one_d = extract_1d()
    matrix = extract_matrix()
    g2, g3 = extract_nd()
    meta = {
        "source": "martinvuyk/hackathon-fft fft/_test_values.mojo, fft/tests.mojo (golden data only)",
        "tolerance": {"atol": 1e-2, "rtol": 1e-5, "source": "fft/tests.mojo:40-41"},
    }
    with open(os.path.join(HERE, "fft_1d.json"), "w") as f:
        json.dump({"meta": meta, "values": one_d, "matrix": matrix}, f, separators=(",", ":"))
    with open(os.path.join(HERE, "fft_2d.json"), "w") as f:
        json.dump({"meta": meta, **g2}, f, separators=(",", ":"))
    with open(os.path.join(HERE, "fft_3d.json"), "w") as f:
        json.dump({"meta": meta, **g3}, f, separators=(",", ":"))
    npairs = sum(len(v) for v in one_d.values())
    print("1-D lengths:", sorted(int(k) for k in one_d), "pairs:", npairs, "matrix rows:", len(matrix))


if __name__ == "__main__":
    main()

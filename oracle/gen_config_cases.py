"""TEST INFRASTRUCTURE ONLY: golden vectors of the config.json reader.  Every text below is loaded by the REFERENCE's own
Config class (oracle/_ref/ref_config_probe = oracle/ref_config_probe.cpp + /root/reference/Code/config.cpp, built by
oracle/Makefile) and queried with its three getters for every key; inputs and the reference's answers go to
tests/golden/config_cases.json.  The texts are this repo's own: the shipped config files of the reference tree plus the corner
cases of its line-oriented grammar (comments inside values, one or three quotes in a key, trailing commas, prefix-parsed
numbers, overflow, wrong types, sections without braces ...).
Run here (needs /root/reference):  make -C oracle ref && python oracle/gen_config_cases.py"""
import glob
import json
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PROBE = os.path.join(HERE, "_ref", "ref_config_probe")
REF = "/root/reference"

CORNERS = [
    '{\n  "settings": {\n    "samples_per_pixel": 4, // trailing comment\n    "max_bounces": 7\n  },\n  "image": {\n    "exposure": 0.25,\n    "shutter_time": 2\n  }\n}\n',
    '"a": 1\n"b": 2.5\n"c": true\n"d": false\n',                                   # no section: "global."
    '"s": {\n"url": "http://x.y/z",\n"frac": 1.5//2\n"n": 12 // 13\n}\n',           # "//" cuts inside values
    '"s": {\n"one"quote": 3\n"k"e"y": 4\nkey": 5\n"open: 6\n}\n',                    # quotes in odd places
    '"s": {\n"a": 12abc,\n"b": 1.5e2x,\n"c": .5,\n"d": 5.,\n"e": -3,\n"f": +4,\n"g": 0x10,\n"h": 1e3,\n"i": 1.e3\n}\n',   # prefix parsing
    '"s": {\n"big": 99999999999,\n"bigd": 1.0e999,\n"neg": -2147483648,\n"emp": ,\n"emp2":\n"sp":   7   ,\n"tab":\t8\t,\n}\n',
    '"s": {\n"t": True,\n"u": true ,\n"v": "true",\n"w": truee,\n"x": false,,\n"y": 1,,\n}\n',
    '"outer": {\n  "inner": {\n    "deep": 1\n  },\n  "after": 2\n}\n',                # nesting: the last header wins
    '"s" : {\n"a" : 1\n}\n"t":{\n"a":2\n}\n"u": { \n"a": 3\n}\n',                     # spacing around ':' and '{'
    '  "s": {   \r\n  "crlf": 1.25,\r\n  "x": 3\r\n}\r\n',                             # CRLF
    '"s": {\n"dup": 1,\n"dup": 2.5,\n"dup2": 2.5,\n"dup2": true\n}\n',                # later entries replace earlier ones, type too
    '"s": {\n"a:b": 1,\n"c": 1:2,\n"d": "x": 3\n}\n',                                 # colons
    '"s": {\n"inf": inf.0,\n"nan": nan.,\n"i": inf,\n"dot": .,\n"m": -.5e-3\n}\n',
    '{"s": {\n"same_line": 1}\n"next": 2\n}\n',
]


def keys_of(text):
    """Every string in quotes, with every section prefix that occurs: a superset of the keys the reference can store."""
    names = set(re.findall(r'"([^"\n]*)"', text)) | set(re.findall(r'"([^\n]*)"', text)) | {m.strip().strip('"') for m in re.findall(r'^([^:\n]*):', text, re.M)}
    names |= {n.split('"')[-1] for n in list(names)} | {"open", "key", "one\"quote", "k\"e\"y", "a:b"}
    sections = {"global"} | set(re.findall(r'"([^"\n]*)"\s*:\s*\{', text))
    out = set()
    for s in sections:
        for n in names:
            if n and not re.search(r"\s", n):
                out.add(s + "." + n)
    return sorted(out)


def probe(text, keys):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False, newline="") as f:
        f.write(text)
        path = f.name
    try:
        q = "".join("%s %s\n" % (t, k) for k in keys for t in "idb")
        p = subprocess.run([PROBE, path], input=q, capture_output=True, text=True, check=True)
    finally:
        os.unlink(path)
    res = {}
    for ln in p.stdout.splitlines():
        if not ln or ln[0] not in "idb" or ln[1] != " ":
            continue   # "Configuration loaded from ..." line
        t, k, a, b = ln.split(" ")
        res.setdefault(k, {})[t] = [a, b]
    return res


def main():
    texts = list(CORNERS)
    for path in sorted(glob.glob(os.path.join(REF, "Code", "config.json")) + glob.glob(os.path.join(REF, "ASCII", "examples", "**", "*.json"), recursive=True)):
        with open(path, newline="") as f:
            texts.append(f.read())
    cases = []
    for text in texts:
        keys = keys_of(text)
        cases.append({"text": text, "answers": probe(text, keys)})
    with open(os.path.join(ROOT, "tests", "golden", "config_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    stored = sum(1 for c in cases for k, v in c["answers"].items() if v["i"][0] == v["i"][1] or v["b"][0] == v["b"][1])
    print("%d texts, %d keys asked, %d of them hold a value in the reference" % (len(cases), sum(len(c["answers"]) for c in cases), stored))


if __name__ == "__main__":
    main()

"""CPU: lvmhelpers.utils.populate_common_params against a fixture dumped from the REFERENCE's own
parser (tests/golden/make_goldens.py common_params -> tests/golden/common_params.json): every flag
with its type, default, choices and action kind, and the namespaces of representative command
lines (the published scripts' flags)."""
import argparse
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "common_params.json")


def _spec(parser):
    out = []
    for a in parser._actions:
        if not a.option_strings or a.dest == "help":
            continue
        out.append(dict(flags=a.option_strings, dest=a.dest, kind=type(a).__name__,
                        type=None if a.type is None else a.type.__name__, default=a.default,
                        choices=None if a.choices is None else list(a.choices), nargs=a.nargs))
    return out


def test_flags_match_the_reference_parser():
    from lvmhelpers.utils import populate_common_params
    gold = json.load(open(GOLD))
    parser = argparse.ArgumentParser()
    assert populate_common_params(parser) is parser or populate_common_params is not None
    got = _spec(parser)
    assert [g["dest"] for g in got] == [g["dest"] for g in gold["flags"]]      # same flags, same order
    for g, r in zip(got, gold["flags"]):
        assert g == r, (g, r)


def test_parsed_namespaces_match_the_reference_parser():
    from lvmhelpers.utils import populate_common_params
    gold = json.load(open(GOLD))
    parser = argparse.ArgumentParser()
    populate_common_params(parser)
    for case in gold["parsed"]:
        assert vars(parser.parse_args(case["argv"])) == case["namespace"], case["argv"]
    for bad in (["--mode", "nope"], ["--normalizer", "entmax"], ["--latent_size", "x"]):
        try:
            parser.parse_args(bad)
        except SystemExit:
            continue
        raise AssertionError(f"{bad} was accepted")

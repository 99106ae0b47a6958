"""YamlConfig against the one output of the reference itself that its repository holds: the
``pprint(yaml_obj.get_config())`` cell of examples/processing/test_reading_yaml.ipynb, for
examples/processing/process_example.yaml (fixtures: tests/golden/make_yaml_golden.py).

The notebook ran on an older revision of the YAML.  Since then (and only there) the file
changed in the salting block (``noise_tag`` is now spelled ``csd_tag``; config.py:77 maps the old
name onto the new one) and in one trigger entry (renamed ``of2x1_shared``, two new keys).
Everything else -- the whole ``feature`` section the hot path is driven by (trace lengths in
samples, comma / ``all`` expansion, disabled algorithms dropped, weights, traces_config,
channel_list order), ``global``, ``didv``, ``noise``, ``template`` and the rest of ``trigger`` /
``salting`` -- must be reproduced exactly."""
import ast
import json
import os

from detprocess_amd import YamlConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHANNELS = ["Melange025pcLeft", "Melange025pcRight", "Melange4pc1ch", "Melange1pc1ch"]   # notebook cell 3


def _load():
    with open(os.path.join(GOLDEN, "yaml_example_input.json")) as fh:
        parsed = json.load(fh)
    with open(os.path.join(GOLDEN, "yaml_example_printed.txt")) as fh:
        printed = ast.literal_eval(fh.read())
    return parsed, printed


def test_feature_section_matches_the_reference_output_exactly():
    parsed, printed = _load()
    got = YamlConfig(parsed, CHANNELS, sample_rate=1.25e6).get_config()
    assert set(got) == set(printed)
    for section in ("feature", "global", "didv", "noise", "template"):
        assert got[section] == printed[section], section
    # list order matters downstream (pprint sorts dict keys, so only lists carry an order)
    assert got["feature"]["channel_list"] == printed["feature"]["channel_list"]
    assert got["feature"]["traces_config"] == {(25000, 12500): printed["feature"]["channel_list"]}


def test_trigger_and_salting_sections_match_up_to_the_yaml_revision():
    parsed, printed = _load()
    got = YamlConfig(parsed, CHANNELS, sample_rate=1.25e6).get_config()
    want_salt = printed["salting"]
    for cc in want_salt["channels"].values():
        cc["csd_tag"] = cc.pop("noise_tag")               # obsolete key, config.py:71-79
    assert got["salting"] == want_salt
    want_trig, got_trig = printed["trigger"], got["trigger"]
    assert got_trig["overall"] == want_trig["overall"]
    assert got_trig["channel_list"] == want_trig["channel_list"]
    renamed = {"of2x1_shared_Melange025pc": "of2x2_shared_Melange025pc"}
    assert {renamed.get(k, k) for k in got_trig["channels"]} == set(want_trig["channels"])
    for key, cc in got_trig["channels"].items():
        want = want_trig["channels"][renamed.get(key, key)]
        if key in renamed:
            assert {k: cc[k] for k in want} == want       # the older entry is a subset
            assert set(cc) - set(want) == {"run_residual", "sat_amps_50kHz"}
        else:
            assert cc == want

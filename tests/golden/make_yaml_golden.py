"""Make the YamlConfig fixtures from the reference's own artefacts (run in the build container,
where /root/reference exists; the fixtures travel, the reference does not):

  yaml_example_input.json    the parsed content of examples/processing/process_example.yaml
                             (input DATA: keys and values, no text of the file)
  yaml_example_printed.txt   the ``pprint(yaml_obj.get_config())`` output stored in
                             examples/processing/test_reading_yaml.ipynb cell 6 -- the one
                             output of the reference itself that the repository holds
                             (SURVEY.md section 4); a Python literal.

The notebook was run on an older revision of the YAML; tests/test_yaml_golden.py lists the
entries that changed since.
"""
import json
import os
import sys

import yaml

REF = "/root/reference/examples/processing"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    with open(os.path.join(REF, "process_example.yaml")) as fh:
        parsed = yaml.safe_load(fh)
    with open(os.path.join(HERE, "yaml_example_input.json"), "w") as fh:
        json.dump(parsed, fh, indent=1, sort_keys=False)
    nb = json.load(open(os.path.join(REF, "test_reading_yaml.ipynb")))
    printed = None
    for cell in nb["cells"]:
        for out in cell.get("outputs", []):
            text = "".join(out.get("text") or out.get("data", {}).get("text/plain") or "")
            if text.startswith("{'didv'"):
                printed = text
    if printed is None:
        sys.exit("printed configuration not found in the notebook")
    with open(os.path.join(HERE, "yaml_example_printed.txt"), "w") as fh:
        fh.write(printed)


if __name__ == "__main__":
    main()

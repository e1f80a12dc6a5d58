#!/usr/bin/env python3
"""Model-config tool: describe a table set + FC widths in JSON, have the library validate it,
and print the placement on MI355X's memory hierarchy and the table-ID shard plan.

  python tools/model_config.py --builtin C --shards 8
  python tools/model_config.py --spec my_model.json --shards 4
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--builtin", choices=["A", "B", "C"])
    ap.add_argument("--spec", help="JSON file, see fleetrec_amd.Model.from_spec")
    ap.add_argument("--shards", type=int, default=1)
    args = ap.parse_args()
    fr = g.load_package()
    if args.spec:
        m = fr.Model.from_spec(json.load(open(args.spec)))
    else:
        m = fr.Model.builtin({"A": fr.MODEL_A, "B": fr.MODEL_B, "C": fr.MODEL_C}[args.builtin or "A"])
    rep = m.placement_report()
    print("model %s: %d tables, %.3f GB, record %d floats (%d dense), FC %s" % (m.name, m.n_tables, rep["table_bytes"] / 1e9, m.record_len,
                                                                              m.dense_len, "-".join(map(str, m.fc))))
    print("placement under uniform access (smallest tables first): %s" % rep["levels"])
    print("gather per item: %d rows, %d useful bytes, %d line bytes beyond L2" % (rep["rows_per_item"], rep["useful_row_bytes_per_item"],
                                                                                rep["line_bytes_per_item_beyond_l2"]))
    if args.shards > 1:
        off, ln, pad = m.shard_plan(args.shards)
        print("table-ID shard plan over %d GPUs (record float ranges): %s, padded slice F=%d" % (
            args.shards, [(o, o + l) for o, l in zip(off, ln)], pad))


if __name__ == "__main__":
    main()

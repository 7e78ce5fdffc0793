"""List or print the tensors of a TensorBundle checkpoint, like TensorFlow's ``inspect_checkpoint`` tool::

    python -m distributed_tensorflow_amd.tools.inspect_checkpoint logs                  # latest checkpoint of a directory
    python -m distributed_tensorflow_amd.tools.inspect_checkpoint logs/model.ckpt-30000 --tensor_name dense_1/bias

Works on the files the savers here write (what ``MonitoredTrainingSession(checkpoint_dir=...)`` leaves behind,
/root/reference/example.py:191) and on any other V2 checkpoint whose tensors are float32 / int64 / int32.
"""
from __future__ import annotations

import argparse
import os
import sys

from ..saver import BundleReader, latest_checkpoint


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("checkpoint", help="checkpoint prefix (model.ckpt-N) or a directory with a `checkpoint` state file")
    ap.add_argument("--tensor_name", default="", help="print this tensor's values instead of the listing")
    ap.add_argument("--all_tensors", action="store_true", help="print the values of every tensor")
    args = ap.parse_args(argv)
    prefix = args.checkpoint
    if os.path.isdir(prefix):
        latest = latest_checkpoint(prefix)
        if latest is None:
            print("no checkpoint state file in %s" % prefix, file=sys.stderr)
            return 1
        prefix = latest
    with BundleReader(prefix) as r:
        if args.tensor_name:
            if not r.has_tensor(args.tensor_name):
                print("tensor %r is not in %s" % (args.tensor_name, prefix), file=sys.stderr)
                return 1
            t = r.get_tensor(args.tensor_name)
            print("tensor_name: %s  shape=%s dtype=%s" % (args.tensor_name, tuple(t.shape), str(t.dtype).replace("torch.", "")))
            print(t.numpy())
            return 0
        total = 0
        print("%s  (%d data shard%s)" % (prefix, r.num_shards, "" if r.num_shards == 1 else "s"))
        for name in sorted(r.entries):
            e = r.entries[name]
            n = 1
            for d in e["shape"]:
                n *= d
            total += n
            print("%-44s %-18s shard %d  %10d bytes" % (name, str(tuple(e["shape"])), e["shard"], e["nbytes"]))
            if args.all_tensors:
                print(r.get_tensor(name).numpy())
        print("%d tensors, %d elements" % (len(r.entries), total))
    return 0


if __name__ == "__main__":
    sys.exit(main())

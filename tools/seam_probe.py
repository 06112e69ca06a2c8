#!/usr/bin/env python3
"""Ask the GPU box whether plain device memory stays coherent across the kernel/stream/copy seams the trainers
rely on (cymf_device_seam_probe).  Prints one JSON object per memory type."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cymf_amd import _lib  # noqa: E402

if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print(json.dumps({"device": _lib.device_name(0)}))
    for mt in (0, 1, 2):
        print(json.dumps(_lib.seam_probe(0, mt, rounds)), flush=True)

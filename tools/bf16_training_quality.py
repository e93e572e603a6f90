"""fp32 vs bf16 training trajectories on the synthetic census set (tests/bf16_quality.py) -> JSON:
    python3 tools/bf16_training_quality.py > gpurun_out/r3/bf16_training.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.bf16_quality import run  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(run(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 200)))

#!/bin/bash
# One gpurun call: one training step per library under build_variants/ (tools/variants.py build src=edge_bwd ...), the two E x H passes' times
export TMPDIR=/tmp
python tools/variants.py run probe=train 2>&1 | python tools/_bwd_knobs_fmt.py

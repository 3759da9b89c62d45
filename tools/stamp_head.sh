#!/bin/bash
# records the commit of the working tree in .head (git-ignored): the gpurun snapshot carries no .git
git rev-parse --short HEAD > "$(git rev-parse --show-toplevel)/.head"

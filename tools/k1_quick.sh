#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do tools/k1_variants.sh "$@" ; done 2>&1 | grep "^\[\|check"

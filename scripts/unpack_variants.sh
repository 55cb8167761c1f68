#!/bin/bash
# restores picha_amd/variants/*.so from the shipped *.so.xz (scripts/build_variant.sh); run first in every GPU session that uses variant builds
for f in picha_amd/variants/*.so.xz; do [ -e "$f" ] && xz -dkf "$f"; done
ls picha_amd/variants/*.so 2>/dev/null | wc -l

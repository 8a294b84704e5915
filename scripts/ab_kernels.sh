#!/bin/bash
# same-box comparison of builds, kernel table only (no verification: experiments may compute garbage on purpose):
#   scripts/ab_kernels.sh name1 name2 ...   ("base" = libako.so, otherwise ako_amd/libako_<name>.so; name:VAR=value adds environment)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
  for A in "$@"; do
    N=${A%%:*}; E=""; [ "$A" != "$N" ] && E=${A#*:}
    L=""; [ "$N" != base ] && L=$R/ako_amd/libako_$N.so
    echo "$A $(env $E AKO_LIB_OVERRIDE=$L TOP=${TOP:-2} python3 $R/scripts/bench_nocheck.py 2>/dev/null)"
  done
done

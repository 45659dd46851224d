#!/bin/bash
# (historical) Same-box A/B of the heads evaluation inside the single-launch search, as run for
# profiles/r02_mfma_heads_ab.txt at commit 0ad1782 ("experiment: MFMA heads"): check that commit out, build
# tools/libsmz_pk.so with -DSMZ_DENSE_PK=1 (the packed even/odd arithmetic, which is what the tree ships again), then
#   SMZ_LIB_PATH=tools/libsmz_pk.so SMZ_SEARCH_MFMA=0 python bench.py ...   # pk
#   SMZ_SEARCH_MFMA=0 python bench.py ...                                    # chain on the vector units
#   SMZ_SEARCH_MFMA=1 python bench.py ...                                    # chain on the matrix cores
echo "see profiles/r02_mfma_heads_ab.txt; the experiment lives at commit 0ad1782"

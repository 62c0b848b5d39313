tools/pmc_shape.sh r03c cin32 "32 32 32 32 3 1 1 res"
tools/pmc_shape.sh r03c pg32 "32 32 32 32 3 1 1 res" PCC_WINO_PER_GROUP=1
tools/pmc_shape.sh r03c cin64 "32 16 64 64 3 1 1 res"
tools/pmc_shape.sh r03c g1 "32 64 16 16 3 1 1 res"

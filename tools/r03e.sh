tools/pmc_shape.sh r03e cin32_nores "32 32 32 32 3 1 1"
tools/pmc_shape.sh r03e cin64_nores "32 16 64 64 3 1 1"
tools/pmc_shape.sh r03e g1_nores "32 64 16 16 3 1 1"

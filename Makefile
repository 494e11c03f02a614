# convenience targets; the driver entry points are __graft_entry__.build()/smoke() and bench.py
.PHONY: build test test-gpu bench clean
build:
	python3 -c "import __graft_entry__ as g; g.build()"
test: build
	python3 -m pytest tests -x -q -m "not gpu"
test-gpu: build
	python3 -m pytest tests -x -q -m gpu
bench: build
	python3 bench.py
clean:
	$(MAKE) -C corona-13_amd clean
	rm -f oracle/liboracle.so

# sanitize: the CPU-side native code (plain C host library, the oracle's restatement) built with -fsanitize=address,undefined into
# build/sanitize/ and the CPU test suite run against those builds (the reference's own debug switch: /root/reference/Makefile:104-110).
# The HIP library has no CPU build and GPU AddressSanitizer is not available on the pool: it is not part of this target.
SAN_DIR := build/sanitize
SAN_FLAGS := -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined
.PHONY: sanitize
sanitize:
	mkdir -p $(SAN_DIR)
	# the QBVH builder and the LUT reader keep the product's floating-point contract (-O3 -ffast-math + FMA, corona-13_amd/Makefile): the tree is compared bit for bit
	gcc -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -O3 -ffast-math -fno-finite-math-only -march=x86-64-v3 -fPIC -std=c11 -Wall -D_GNU_SOURCE \
	  -Iinclude -Icorona-13_amd/host -c corona-13_amd/host/ch_qbvh.c -o $(SAN_DIR)/ch_qbvh.o
	gcc -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -O3 -ffast-math -fno-finite-math-only -march=x86-64-v3 -fPIC -std=c11 -Wall -D_GNU_SOURCE \
	  -Iinclude -Icorona-13_amd/host -c corona-13_amd/host/ch_rgb2spec_lut.c -o $(SAN_DIR)/ch_rgb2spec_lut.o
	gcc $(SAN_FLAGS) -fPIC -std=c11 -Wall -Wno-format-truncation -D_GNU_SOURCE -Iinclude -Icorona-13_amd/host -shared \
	  corona-13_amd/host/ch_scene.c corona-13_amd/host/ch_rgb2spec.c corona-13_amd/host/ch_pfm.c $(SAN_DIR)/ch_qbvh.o $(SAN_DIR)/ch_rgb2spec_lut.o \
	  -o $(SAN_DIR)/libcorona_host.so -lm -ldl
	gcc $(SAN_FLAGS) -fPIC -std=c11 -Wall -Wno-unused-function -fno-strict-aliasing -D_GNU_SOURCE -Iinclude -pthread -shared oracle/*.c -o $(SAN_DIR)/liboracle.so -lm
	LD_PRELOAD="$$(gcc -print-file-name=libasan.so) $$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	  CORONA_HOST_LIB=$(CURDIR)/$(SAN_DIR)/libcorona_host.so CORONA_ORACLE_LIB=$(CURDIR)/$(SAN_DIR)/liboracle.so \
	  python3 -m pytest tests/test_host.py tests/test_oracle_golden.py tests/test_multigpu_cpu.py -x -q -m "not gpu" -p no:cacheprovider

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

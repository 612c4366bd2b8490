# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
.PHONY: build test test-gpu bench node clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
node:
	$(MAKE) -C bindings/node
clean:
	$(MAKE) -C sublinear_time_solver_amd/csrc clean
	$(MAKE) -C oracle clean 2>/dev/null || true
	$(MAKE) -C bindings/node clean

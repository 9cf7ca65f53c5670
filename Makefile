# Builds everything in-tree: the device library (gfx950, hipcc cross-compiles without a GPU),
# the C++ host with its tools, and the CPU checker used by the tests.
.PHONY: all lib host checker test clean
all: lib host checker
lib:
	$(MAKE) -C raytracer-rust_amd/csrc
host: lib
	$(MAKE) -C host
checker:
	$(MAKE) -C oracle
test: all
	python -m pytest tests -x -q -m "not gpu"
clean:
	$(MAKE) -C raytracer-rust_amd/csrc clean
	$(MAKE) -C host clean
	$(MAKE) -C oracle clean

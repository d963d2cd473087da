# Build the native pieces without Python (what limitador_amd/build.py and __graft_entry__.build() do).
#   make            librl_engine.so (hipcc, gfx950), librl_storage.so, librl_sharded.so, the C oracle
#   make harness    tests/cpp/abi_harness (C++ host of the C ABI, checked against the oracle; needs a MI355X to run)
ROCM ?= /opt/rocm
HIPCC ?= $(ROCM)/bin/hipcc
LIB ?= limitador_amd/lib
CSRC := limitador_amd/csrc
# make LIB=limitador_amd/lib/exp DEFS=-DRL_EXPERIMENT   the build the test-suite loads (every switch readable)
DEFS ?=

all: $(LIB)/librl_engine.so $(LIB)/librl_storage.so $(LIB)/librl_sharded.so oracle/liblimitador_oracle.so

$(LIB)/librl_engine.so: $(wildcard $(CSRC)/*.hip $(CSRC)/*.hpp) include/rl_engine.h
	mkdir -p $(LIB)
	$(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $(DEFS) -Iinclude $(CSRC)/rl_engine.hip -o $@

$(LIB)/librl_storage.so: $(LIB)/librl_engine.so $(wildcard $(CSRC)/host/*.cpp $(CSRC)/host/*.hpp) include/rl_storage.h include/rl_ingest.h
	g++ -O2 -std=c++17 -fPIC -shared -pthread $(DEFS) -Iinclude $(CSRC)/host/gpu_counter_storage.cpp $(CSRC)/host/ingest.cpp -o $@ \
	    -L$(LIB) -lrl_engine '-Wl,-rpath,$$ORIGIN'

$(LIB)/librl_sharded.so: $(LIB)/librl_engine.so $(CSRC)/host/rl_sharded.cpp include/rl_sharded.h
	g++ -O2 -std=c++17 -fPIC -shared -pthread -D__HIP_PLATFORM_AMD__ $(DEFS) -Iinclude -I$(ROCM)/include $(CSRC)/host/rl_sharded.cpp -o $@ \
	    -L$(LIB) -lrl_engine -L$(ROCM)/lib -lamdhip64 -ldl '-Wl,-rpath,$$ORIGIN' -Wl,-rpath,$(ROCM)/lib

oracle/liblimitador_oracle.so: oracle/limitador_oracle.c oracle/limitador_oracle.h
	$(MAKE) -C oracle

harness: all
	g++ -O2 -std=c++17 -Wall tests/cpp/abi_harness.cpp -Iinclude -Ioracle -L$(LIB) -lrl_engine -Loracle -llimitador_oracle \
	    -Wl,-rpath,$(CURDIR)/$(LIB) -Wl,-rpath,$(CURDIR)/oracle -o tests/cpp/abi_harness

clean:
	rm -f $(LIB)/*.so oracle/*.so tests/cpp/abi_harness

.PHONY: all harness clean

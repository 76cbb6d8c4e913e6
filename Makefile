# Top-level developer targets (test infrastructure; the product's own build is meters.lv2_amd/csrc/Makefile,
# which __graft_entry__.build() drives).
#
#   make build        the product + the parity checkers (= __graft_entry__.build())
#   make test         the CPU test suite
#   make check-asan   the host C under AddressSanitizer + UndefinedBehaviorSanitizer: lv2_*.c (the LV2
#                     glue), mtr_setup.c (set-up math; inside an engine library whose HIP objects are the normal ones)
#                     and oracle/mtr_oracle.c are rebuilt with -fsanitize=address,undefined into meters.lv2_amd/lib_asan
#                     and the CPU suite runs against them (MTR_LIB / MTR_PLUGIN_SO / MTR_ORACLE_SO point the tests at
#                     the instrumented libraries; the sanitizer runtimes are preloaded because the host process is
#                     python; leak checking is off: the interpreter's own allocations drown it).
CC      ?= gcc
ROOT    := $(abspath .)
CSRC    := meters.lv2_amd/csrc
OBJ     := meters.lv2_amd/lib/obj
ASAN    := meters.lv2_amd/lib_asan
INC     := -I$(ROOT)/include -I$(CSRC)
LV2SRC  := $(addprefix $(CSRC)/,lv2_plugin.c lv2_ebur128.c lv2_intstat.c lv2_needle.c lv2_dr14.c)
SANFLAGS := -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined

.PHONY: build test check-asan
build:
	python3 -c "import __graft_entry__ as g; g.build()"

test:
	python3 -m pytest tests -x -q -m "not gpu"

check-asan: build
	@mkdir -p $(ASAN)/obj
	$(CC) $(SANFLAGS) -std=gnu11 -fPIC -ffp-contract=off -msse2 -mfpmath=sse $(INC) -Wall -c $(CSRC)/mtr_setup.c -o $(ASAN)/obj/mtr_setup.o
	g++ -shared -fPIC -fsanitize=address,undefined -o $(ASAN)/libmtr_engine.so $(filter-out $(OBJ)/mtr_setup.o,$(wildcard $(OBJ)/*.o)) \
	    $(ASAN)/obj/mtr_setup.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lrccl -lm
	$(CC) $(SANFLAGS) -std=gnu11 -fPIC -shared -fvisibility=hidden -ffp-contract=off -msse2 -mfpmath=sse $(INC) -Wall \
	    -o $(ASAN)/meters_amd.so $(LV2SRC) -L$(ASAN) -lmtr_engine -Wl,-rpath,'$$ORIGIN' -lm
	$(CC) $(SANFLAGS) -msse2 -mfpmath=sse -ffp-contract=off -std=gnu11 -fPIC -Wall -Wno-unused-function -shared \
	    -o $(ASAN)/libmtr_oracle.so oracle/mtr_oracle.c -lm
	LD_PRELOAD="$$($(CC) -print-file-name=libasan.so) $$($(CC) -print-file-name=libubsan.so)" \
	    ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	    MTR_LIB=$(ROOT)/$(ASAN)/libmtr_engine.so MTR_PLUGIN_SO=$(ROOT)/$(ASAN)/meters_amd.so \
	    MTR_ORACLE_SO=$(ROOT)/$(ASAN)/libmtr_oracle.so \
	    python3 -m pytest tests -x -q -m "not gpu and not ref" -p no:cacheprovider

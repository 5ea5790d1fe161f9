# builds A/B variants of the hand-scheduled kernel into lambdaworks_kzg_amd/lib_<name>/ (git-ignored; they travel to the GPU box)
set -e
cd "$(dirname "$0")/.."
build() {  # name, env...
  name=$1; shift
  env "$@" python3 tools/gen_direct_asm.py > /dev/null
  make -s -C lambdaworks_kzg_amd/csrc -j8 OUT_DIR=../lib_$name OBJ_DIR=../build_$name 2>&1 | grep -v hipcc | tail -3
}
for v in "$@"; do
  case $v in
    sdst) build sdst LWK_ASM_SDST=1 ;;
    block2) build block2 LWK_ASM_BLOCK=2 ;;
    block4) build block4 LWK_ASM_BLOCK=4 ;;
    serial) build serial LWK_ASM_BLOCK=100000 ;;
  esac
done
python3 tools/gen_direct_asm.py > /dev/null   # back to the committed stream
python3 tools/gen_direct_asm.py --check

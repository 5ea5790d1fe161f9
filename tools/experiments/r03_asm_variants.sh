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
    confine) build confine LWK_ASM_CONFINE=1 ;;
    confine2) build confine2 LWK_ASM_CONFINE=2 ;;
    confine3) build confine3 LWK_ASM_CONFINE=3 ;;
    rotwin) build rotwin LWK_ASM_ROTWIN=1 ;;
    nop_top2) build nop_top2 LWK_ASM_PAUSE=nop_top2 ;;
    nop_top_half) build nop_top_half LWK_ASM_PAUSE=nop_top_half ;;
    nop_mid) build nop_mid LWK_ASM_PAUSE=nop_mid ;;
    nop0) build nop0 LWK_ASM_PAUSE=nop0 ;;
    align64) build align64 LWK_ASM_PAUSE=align64 ;;
    align64_nop0) build align64_nop0 LWK_ASM_PAUSE=align64_nop0 ;;
    e64) build e64 LWK_ASM_E64=1 ;;
    nop_top) build nop_top LWK_ASM_PAUSE=nop_top ;;
    sleep_top) build sleep_top LWK_ASM_PAUSE=sleep_top ;;
    nop_groups) build nop_groups LWK_ASM_PAUSE=nop_groups ;;
  esac
done
python3 tools/gen_direct_asm.py > /dev/null   # back to the committed stream
python3 tools/gen_direct_asm.py --check

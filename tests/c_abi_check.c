/* The public header is a C header: compiled as strict C99, linked against libpwicp.so, run (tests/test_abi.py). */
#include "pwicp.h"
#include <stdio.h>
int main(void) {
    pwicp_context* ctx = 0;
    int rc = pwicp_create(&ctx, 0);
    printf("pwicp_create -> %d (%s), %s, result struct %zu B, record %zu B\n", rc, rc ? "no device" : "ok", pwicp_version(),
           sizeof(pwicp_result), sizeof(pwicp_pair_record));
    if (!rc) pwicp_destroy(ctx);
    return 0;
}

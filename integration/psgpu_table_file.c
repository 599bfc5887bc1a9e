/* integration/psgpu_table_file.c -- see psgpu_table_file.h */
#include <string.h>
#include "psgpu_table_file.h"

FILE *
psgpu_table_file_open(const char *path)
{
    FILE *fp = fopen(path, "wb");
    if (fp) fwrite("PSGB1\n", 1, 6, fp);
    return fp;
}

void
psgpu_table_file_put(void *ctx, const char *name, char dt, int nd, const int64_t *dims, const void *data)
{
    FILE *fp = (FILE *)ctx;
    uint32_t nl = (uint32_t)strlen(name), d = (uint32_t)dt, n_d = (uint32_t)nd;
    size_t esz = (dt == 'f' || dt == 'i') ? 4 : (dt == 'h' || dt == 'H') ? 2 : (dt == 'q' || dt == 'd') ? 8 : 1;
    size_t n = 1;
    int i;
    for (i = 0; i < nd; ++i) n *= (size_t)dims[i];
    fwrite(&nl, 4, 1, fp); fwrite(name, 1, nl, fp);
    fwrite(&d, 4, 1, fp); fwrite(&n_d, 4, 1, fp);
    fwrite(dims, 8, nd, fp);
    if (n) fwrite(data, esz, n, fp);
}

int
psgpu_table_file_close(FILE *fp)
{
    int bad = ferror(fp);
    return (fclose(fp) != 0 || bad) ? -1 : 0;
}

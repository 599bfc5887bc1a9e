/* psgpu_table_file.h -- the table file: a flat container of named arrays ("PSGB1": magic line, then per array name length,
 * name, type letter, rank, dims (int64), data; little endian), written by reference-side tools out of a live decoder and
 * read back by pocketsphinx_amd/tablefile.py.  How a task's search tables and language model reach the Python product and
 * bench.py without a reference decoder in the process (integration/psgpu_export_tables.c). */
#ifndef PSGPU_TABLE_FILE_H
#define PSGPU_TABLE_FILE_H
#include <stdint.h>
#include <stdio.h>
#ifdef __cplusplus
extern "C" {
#endif
FILE *psgpu_table_file_open(const char *path);          /* NULL: could not be created */
/* dt: 'f' float32, 'i' int32, 'h' int16, 'B' uint8, 'H' uint16, 'q' int64, 'd' float64.  The signature is
 * psgpu_table_emit_fn's (psgpu_search_tables.h) with ctx = the FILE. */
void psgpu_table_file_put(void *fp, const char *name, char dt, int nd, const int64_t *dims, const void *data);
int psgpu_table_file_close(FILE *fp);                   /* 0, or -1 when a write failed */
#ifdef __cplusplus
}
#endif
#endif

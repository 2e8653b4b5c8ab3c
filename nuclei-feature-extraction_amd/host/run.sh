#!/bin/bash
# Folder driver for the extraction binary -- the counterpart of the reference's run.sh (run.sh:43-55):
# every <name>.geojson of a folder is processed against the slide of the same name and written to
# <output_folder>/<name>.<ext>.  Same arguments and the same two environment variables:
#   EXTRACTION_ARGS      options placed before the positionals (e.g. "-b 100 -p 64 -g 0 -g 1")
#   EXTRACTION_FEATURES  feature sets placed after them (e.g. "geometry color")
# The reference always pairs a .svs slide; this build reads PNG tiles (no OpenSlide in the image), so the slide
# extension is looked up in SLIDE_EXTENSIONS (default "svs png", first existing file wins).
set -u
usage() {
    echo "Usage: $0 <geojson_folder> <slide_folder> <output_folder> [output_file_extension]"
    echo "  <geojson_folder>        Folder containing geojson files"
    echo "  <slide_folder>          Folder containing slide files"
    echo "  <output_folder>         Folder to store output files"
    echo "  [output_file_extension] File extension of output files (default: csv)"
    exit 1
}
[ $# -ge 3 ] && [ -n "$1" ] && [ -n "$2" ] && [ -n "$3" ] || usage
geojson_folder=$1; slide_folder=$2; output_folder=$3; ext=${4:-csv}
here=$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)
binary=${NFX_BINARY:-$here/nuclei-feature-extraction}
mkdir -p "$output_folder"
shopt -s nullglob
status=0
for geojson in "$geojson_folder"/*.geojson; do
    name=$(basename "$geojson" .geojson)
    slide=$slide_folder/$name.svs
    for e in ${SLIDE_EXTENSIONS:-svs png}; do
        if [ -f "$slide_folder/$name.$e" ]; then slide=$slide_folder/$name.$e; break; fi
    done
    echo "Processing $geojson"
    # shellcheck disable=SC2086  (the two variables are word lists by contract)
    if "$binary" ${EXTRACTION_ARGS:-} "$geojson" "$slide" "$output_folder/$name.$ext" ${EXTRACTION_FEATURES:-}; then
        echo "Done processing $geojson"
    else
        echo "Error processing $geojson"
        status=1
    fi
done
exit $status

"""Cuts the first 400 records of the reference's MEG test recording
(/root/reference/test_data/meg/subj01_1ksamples.tfrecords, a data file of the reference's
own tests: BUILD.bazel:40-47) into tests/golden/meg_subj01_400.tfrecords, keeping only the
'meg' and 'envelope' features.  Run in the build container (the reference is not on the GPU
box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from telluride_decoding_amd import tfrecord

src = '/root/reference/test_data/meg/subj01_1ksamples.tfrecords'
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'meg_subj01_400.tfrecords')
feats = tfrecord.read_file(src, fields=['meg', 'envelope'], verify=True)
tfrecord.write_file(dst, {k: v[:400] for k, v in feats.items()})
print(dst, os.path.getsize(dst))

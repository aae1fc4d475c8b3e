"""Recipe plumbing around the fused Conv-TasNet path: wav I/O without torchaudio,
the wsj0-mix style datasets / loaders, a trainer that writes and reads the reference's checkpoint format, and a tester
for variable-length utterances.  Counterparts in the reference: egs/wsj0-mix/common/src/{dataset,driver}.py and
egs/wsj0-mix/conv-tasnet/local/{train,test}.py."""

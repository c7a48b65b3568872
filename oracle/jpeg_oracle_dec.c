/* decoder restatement: added below */
